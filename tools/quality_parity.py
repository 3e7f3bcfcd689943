"""GPU-box experiment: ranking-quality parity (hit_rate@10 and friends, factor norms) between the Hogwild engine and the
sequential CPU oracle (the pinned restatement of the reference) on the planted MovieLens-1M-shaped surrogate, several seeds.
BASELINE.json config 1: factors=20, loss='bpr', epochs=5.   (uses oracle/: tooling, not product)"""
import argparse
import os
import sys
import time

import numpy as np
import pandas as pd

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from _gpulock import gpu
from oracle import oracle as orc
from rankfm_amd import EngineOptions, RankFM, evaluation, synthetic

ap = argparse.ArgumentParser()
ap.add_argument("--users", type=int, default=6040)
ap.add_argument("--items", type=int, default=3706)
ap.add_argument("--factors", type=int, default=20)
ap.add_argument("--loss", default="bpr")
ap.add_argument("--max-samples", type=int, default=20)
ap.add_argument("--epochs", type=int, default=5)
ap.add_argument("--seeds", type=int, default=3)
ap.add_argument("--tags", type=int, default=0)
ap.add_argument("--schedule", default="constant")
ap.add_argument("--workgroups", default="0")
ap.add_argument("--dampings", default="0")
ap.add_argument("--variants", default="", help="engine variants for the GPU sides, ';'-separated; a variant is a ','-separated list of "
                "geometry overrides (rfm_fit_tuning): e.g. ';segment_rows=16'")
ap.add_argument("--seed0", type=int, default=0)
ap.add_argument("--large", action="store_true", help="synthetic.make_planted_large (config-2-sized problems) instead of make_planted")
ap.add_argument("--degree", type=float, default=60.0, help="mean degree of make_planted_large")
ap.add_argument("--metrics", default="hit,mrr,prec,rec")
a = ap.parse_args()

rows = []
T0 = time.time()


def engine_of(variant, **kw):
    tune = {x.split("=")[0]: int(x.split("=")[1]) for x in variant.split(",") if "=" in x}
    return EngineOptions(tune=tune, **kw)


for seed in range(a.seed0, a.seed0 + a.seeds):
    d = (synthetic.make_planted_large(a.users, a.items, seed=seed, mean_degree=a.degree) if a.large
         else synthetic.make_planted(a.users, a.items, seed=seed, n_tags=a.tags))
    train, test = pd.DataFrame(d["train"], columns=["u", "i"]), pd.DataFrame(d["test"], columns=["u", "i"])
    print("seed %d: data ready, %d train rows (%.0f s since start)" % (seed, len(train), time.time() - T0), flush=True)
    uf = itf = None
    if a.tags:
        uf = pd.concat([pd.DataFrame({"u": np.arange(a.users)}), pd.DataFrame(d["user_tags"])], axis=1)
        itf = pd.concat([pd.DataFrame({"i": np.arange(a.items)}), pd.DataFrame(d["item_tags"])], axis=1)
    res = {}
    sides = ["oracle"] + ["gpu:%s:%s:%s" % (w, m, e) for w in a.workgroups.split(",") for m in a.dampings.split(",") for e in a.variants.split(";")]
    for side in sides:
        wg, damp = (int(side.split(":")[1]), float(side.split(":")[2])) if side != "oracle" else (0, 0.0)
        m = RankFM(factors=a.factors, loss=a.loss, max_samples=a.max_samples, learning_schedule=a.schedule,
                   engine=engine_of(side.split(":", 3)[3] if side != "oracle" else "", seed=100 + seed, n_workgroups=wg, damping=damp))
        np.random.seed(seed)
        t0 = time.time()
        if side != "oracle":
            with gpu():
                m.fit(train, uf, itf, epochs=a.epochs)
        else:
            m._init_all(train, uf, itf, None)          # same initial weights (same numpy stream) as the GPU side
            ms = 1 if a.loss == "bpr" else a.max_samples
            orc.fit(m.interactions, m.sample_weight, m.user_items.offsets, m.user_items.items, m.x_uf, m.x_if, m.w_i, m.w_if,
                    m.v_u, m.v_i, m.v_uf, m.v_if, m.alpha, m.beta, m.learning_rate, m.learning_schedule, m.learning_exponent,
                    ms, a.epochs, perms=None, rng_mode=orc.RNG_COUNTER, seed=100 + seed, membership="binary")
            m.is_fit = True
        dt = time.time() - t0
        fns = dict(hit=evaluation.hit_rate, mrr=evaluation.reciprocal_rank, prec=evaluation.precision, rec=evaluation.recall)
        res[side] = dict({k: fns[k](m, test, k=10) for k in a.metrics.split(",")},
                         nvu=np.linalg.norm(m.v_u), nvi=np.linalg.norm(m.v_i), nwi=np.linalg.norm(m.w_i), t=dt)
        print("  seed %d side %-32s %s   (%.0f s since start)" % (seed, side, {k: round(float(v), 4) for k, v in res[side].items()}, time.time() - T0), flush=True)
    rows.append(res)
    print("seed %d  n_train %d" % (seed, len(train)), {s: {k: round(float(v), 4) for k, v in r.items()} for s, r in res.items()}, flush=True)
for side in [s for s in rows[0] if s != "oracle"]:
    print("==", side)
    for k in a.metrics.split(",") + ["nvu", "nvi", "nwi", "t"]:
        o = np.array([r["oracle"][k] for r in rows]); g = np.array([r[side][k] for r in rows])
        print("%-5s oracle %.4f +- %.4f   gpu %.4f +- %.4f   diff %+.4f (%+.2f%%)" % (k, o.mean(), o.std(), g.mean(), g.std(), g.mean() - o.mean(),
              100 * (g.mean() - o.mean()) / o.mean()))
