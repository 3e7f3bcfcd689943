"""BASELINE config 1 (MovieLens-1M, factors=20, loss='bpr', epochs=5): hit_rate@10 and friends of the MI355X engine next to the
sequential CPU oracle (the pinned restatement of the reference) on the SAME data and initial weights, several seeds.

    RANKFM_ML1M=/path/to/ml-1m/ratings.dat python tools/movielens_quality.py        # real data
    python tools/movielens_quality.py                                                # planted MovieLens-1M-shaped surrogate

The north star's bar is hit_rate@10 within 1 point of the reference.  (uses oracle/: tooling, not product)"""
import os
import sys

import numpy as np
import pandas as pd

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import oracle as orc
from rankfm_amd import EngineOptions, RankFM, datasets, evaluation, synthetic

seeds = int(sys.argv[1]) if len(sys.argv) > 1 else 3
real = datasets.load_movielens_1m()
print("data:", "MovieLens-1M from %s" % real["path"] if real else "planted MovieLens-1M-shaped surrogate (no ratings.dat supplied)")
rows = []
for seed in range(seeds):
    if real:
        train, test = real["train"], real["test"]
    else:
        d = synthetic.make_planted(seed=seed)
        train, test = pd.DataFrame(d["train"], columns=["user_id", "item_id"]), pd.DataFrame(d["test"], columns=["user_id", "item_id"])
    res = {}
    for side in ("oracle", "gpu"):
        m = RankFM(factors=20, loss="bpr", engine=EngineOptions(seed=100 + seed))
        np.random.seed(seed)
        if side == "gpu":
            m.fit(train, epochs=5)
        else:
            m._init_all(train)
            orc.fit(m.interactions, m.sample_weight, m.user_items.offsets, m.user_items.items, m.x_uf, m.x_if, m.w_i, m.w_if, m.v_u, m.v_i,
                    m.v_uf, m.v_if, m.alpha, m.beta, m.learning_rate, "constant", 0.25, 1, 5, perms=None, rng_mode=orc.RNG_COUNTER,
                    seed=100 + seed, membership="binary")
            m.is_fit = True
        res[side] = [evaluation.hit_rate(m, test, k=10), evaluation.reciprocal_rank(m, test, k=10), evaluation.precision(m, test, k=10),
                     evaluation.recall(m, test, k=10), np.linalg.norm(m.v_u), np.linalg.norm(m.v_i), np.linalg.norm(m.w_i)]
    rows.append(res)
    print("seed", seed, {k: np.round(v, 4).tolist() for k, v in res.items()}, flush=True)
o, g = np.array([r["oracle"] for r in rows]), np.array([r["gpu"] for r in rows])
for k, name in enumerate(("hit_rate@10", "mrr@10", "precision@10", "recall@10", "|v_u|", "|v_i|", "|w_i|")):
    print("%-13s oracle %.4f +- %.4f   gpu %.4f +- %.4f   diff %+.4f" % (name, o[:, k].mean(), o[:, k].std(), g[:, k].mean(), g[:, k].std(),
                                                                      g[:, k].mean() - o[:, k].mean()))
