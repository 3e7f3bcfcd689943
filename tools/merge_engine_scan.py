"""GPU-box experiment: the merge rule's constants with the REAL engine in every shard (distributed.emulate_ranks_on_one_device): eight user
shards of a config-2-shaped planted problem, hit_rate@10 / norms against the single-GPU engine on the whole data and the sequential
oracle.  Measurement tooling (uses oracle/), not product.

    python tools/merge_engine_scan.py --variants "0.1:0.3:1;0.03:0.3:1;0.01:0.3:1;0.1:0.3:4"      (c_factors : c_biases : exchanges per epoch)"""
import argparse
import multiprocessing as mp
import os
import sys
import time

import numpy as np
import pandas as pd

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--variants", default="0.1:0.3:1;0.03:0.3:1;0.01:0.3:1;0.003:0.3:1;0.1:0.1:1;0.01:0.1:1")
    ap.add_argument("--seeds", type=int, default=1)
    ap.add_argument("--factors", type=int, default=32)
    ap.add_argument("--epochs", type=int, default=5)
    ap.add_argument("--world", type=int, default=8)
    ap.add_argument("--learning-rate", type=float, default=0.1)
    a = ap.parse_args()
    import torch
    from oracle.planted_worker import fit_pairs
    from rankfm_amd import EngineOptions, RankFM, evaluation, synthetic
    from rankfm_amd.distributed import emulate_ranks_on_one_device
    t0 = time.time()
    data = {s: synthetic.make_planted_large_device(100_000, 50_000, seed=s) for s in range(a.seeds)}
    pool = mp.get_context("spawn").Pool(a.seeds)
    pending = {s: pool.apply_async(fit_pairs, (("o", s, data[s]["train"], a.factors, a.epochs, "bpr", 1),)) for s in data}
    res = {}

    def score(tag, s, weights, train, test):
        o = RankFM(factors=a.factors, loss="bpr", engine=EngineOptions(seed=100 + s))
        np.random.seed(s)
        o._init_all(train)
        for k, v in weights.items():
            setattr(o, k, np.ascontiguousarray(v))
        o.is_fit = True
        res.setdefault(tag, []).append([evaluation.hit_rate(o, test, k=10), np.linalg.norm(o.v_u), np.linalg.norm(o.v_i), np.linalg.norm(o.w_i)])

    for s, d in data.items():
        train, test = pd.DataFrame(d["train"], columns=["u", "i"]), pd.DataFrame(d["test"], columns=["u", "i"])
        m = RankFM(factors=a.factors, loss="bpr", learning_rate=a.learning_rate, engine=EngineOptions(seed=100 + s))
        np.random.seed(s)
        m.fit(train, epochs=a.epochs)
        res.setdefault("one engine, whole data", []).append([evaluation.hit_rate(m, test, k=10), np.linalg.norm(m.v_u), np.linalg.norm(m.v_i), np.linalg.norm(m.w_i)])
        m0 = RankFM(factors=a.factors, loss="bpr", learning_rate=a.learning_rate, engine=EngineOptions(seed=100 + s))
        np.random.seed(s)
        m0._init_all(train)
        problem = dict(interactions=m0.interactions, sample_weight=m0.sample_weight, csr_offsets=m0.user_items.offsets, csr_items=m0.user_items.items,
                       x_uf=m0.x_uf, x_if=m0.x_if, weights={k: np.array(getattr(m0, k), copy=True) for k in ("w_i", "w_if", "v_u", "v_i", "v_uf", "v_if")})
        hyper = dict(alpha=m0.alpha, beta=m0.beta, learning_rate=a.learning_rate, learning_schedule="constant", learning_exponent=0.25, max_samples=1)
        for v in a.variants.split(";"):
            cf, cb, k = v.split(":")
            out = emulate_ranks_on_one_device(problem, a.world, hyper, a.epochs, torch.device("cuda", 0), syncs_per_epoch=(k if k == "auto" else int(k)), seed=100 + s,
                                              c_factors=float(cf), c_biases=float(cb))
            score("%d shards c_factors %s c_biases %s, %s exchange(s)/epoch" % (a.world, cf, cb, k), s, out, train, test)
        if a.learning_rate == 0.1:
            score("sequential oracle", s, pending[s].get(timeout=3000)["weights"], train, test)
    pool.terminate()
    base = np.mean(res["one engine, whole data"], axis=0)
    for tag, rows in res.items():
        r = np.mean(rows, axis=0)
        print("%-64s hit_rate@10 %.4f (%+.2f pt vs one engine)  norms / one engine - 1: v_u %+.1f%% v_i %+.1f%% w_i %+.1f%%"
              % (tag, r[0], 100 * (r[0] - base[0]), 100 * (r[1] / base[1] - 1), 100 * (r[2] / base[2] - 1), 100 * (r[3] / base[3] - 1)), flush=True)
    print("total %.0f s" % (time.time() - t0))


if __name__ == "__main__":
    main()
