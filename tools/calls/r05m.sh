#!/bin/bash
cd ${GRAFT_REPO_ROOT:-.}
O=gpurun_out/r05m; mkdir -p $O
cat > /tmp/late_scan.py <<'PY'
import sys, os
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tests"))
import numpy as np, pandas as pd, torch, multiprocessing as mp
from oracle.planted_worker import fit_pairs
from rankfm_amd import synthetic, EngineOptions, RankFM, evaluation
from rankfm_amd.distributed import emulate_ranks_on_one_device
if __name__ == "__main__":
    seeds = (0, 1)
    data = {s: synthetic.make_planted_large_device(100_000, 50_000, seed=s, n_tags=8) for s in seeds}
    pool = mp.get_context("spawn").Pool(2)
    pend = {s: pool.apply_async(fit_pairs, (("bpr_k32", s, data[s]["train"], 32, 5, "bpr", 1),)) for s in seeds}
    res = {}
    for name, syncs, late in (("block8", 8, False), ("late8", 8, True), ("late12", 12, True), ("late16", 16, True), ("late24", 24, True)):
        hits = []
        for s in seeds:
            d = data[s]
            train, test = pd.DataFrame(d["train"], columns=["u", "i"]), pd.DataFrame(d["test"], columns=["u", "i"])
            m = RankFM(factors=32, loss="bpr", engine=EngineOptions(seed=100 + s)); np.random.seed(s); m._init_all(train)
            problem = dict(interactions=m.interactions, sample_weight=m.sample_weight, csr_offsets=m.user_items.offsets, csr_items=m.user_items.items,
                           x_uf=m.x_uf, x_if=m.x_if, weights={k: getattr(m, k) for k in ("w_i", "w_if", "v_u", "v_i", "v_uf", "v_if")})
            hyper = dict(alpha=m.alpha, beta=m.beta, learning_rate=m.learning_rate, learning_schedule=m.learning_schedule, learning_exponent=m.learning_exponent, max_samples=1)
            out = emulate_ranks_on_one_device(problem, 8, hyper, 5, torch.device("cuda", 0), syncs_per_epoch=syncs, seed=100 + s, late=late)
            o = RankFM(factors=32, loss="bpr", engine=EngineOptions(seed=100 + s)); np.random.seed(s); o._init_all(train)
            for k, v in out.items(): setattr(o, k, np.ascontiguousarray(v))
            o.is_fit = True
            hits.append(evaluation.hit_rate(o, test, k=10))
        res[name] = hits
        print(name, np.round(hits, 4).tolist(), "mean %.4f" % np.mean(hits), flush=True)
    orc = []
    for s in seeds:
        d = data[s]; train, test = pd.DataFrame(d["train"], columns=["u", "i"]), pd.DataFrame(d["test"], columns=["u", "i"])
        o = RankFM(factors=32, loss="bpr", engine=EngineOptions(seed=100 + s)); np.random.seed(s); o._init_all(train)
        for k, v in pend[s].get(timeout=1500)["weights"].items(): setattr(o, k, np.ascontiguousarray(v))
        o.is_fit = True; orc.append(evaluation.hit_rate(o, test, k=10))
    print("oracle", np.round(orc, 4).tolist(), "mean %.4f" % np.mean(orc))
    pool.terminate()
PY
( timeout 1500 python /tmp/late_scan.py ) > $O/late_scan.log 2>&1; tail -8 $O/late_scan.log
