"""GPU-box experiment: where does the engine's ranking-quality gap at config 2's shape come from -- asynchrony, or the ORDER in which it
visits an epoch (user segments of <= 32 consecutive rows in a keyed order, instead of the reference's row-level shuffle)?

Planted problems of config 2's shape (synthetic.make_planted_large_device), a few seeds; per seed
  * the SEQUENTIAL oracle (reference sampler) in a row-level keyed shuffle, and in the engine's segment order with segments of
    <= 32 / 16 / 8 rows (rankfm_amd.order.epoch_positions) -- no asynchrony anywhere: pure order effects;
  * the engine (default; tune segment_rows = 16 / 8; half the workgroups).
hit_rate@10 of every model is evaluated with the GPU recommender.  Test / measurement tooling (uses oracle/), not product.

    python tools/order_quality.py [--seeds 3] [--factors 32] [--epochs 5]"""
import argparse
import multiprocessing as mp
import os
import sys
import time

import numpy as np
import pandas as pd

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
WEIGHTS = ("w_i", "w_if", "v_u", "v_i", "v_uf", "v_if")


def oracle_job(args):
    tag, seed, train, factors, epochs, seg_rows = args
    from oracle import oracle as orc
    from rankfm_amd import EngineOptions, RankFM, order
    orc.build()
    m = RankFM(factors=factors, loss="bpr", engine=EngineOptions(seed=100 + seed))
    np.random.seed(seed)
    m._init_all(pd.DataFrame(train, columns=["u", "i"]))
    pairs, sw, perms = m.interactions, m.sample_weight, None
    if seg_rows:
        by = np.lexsort((pairs[:, 1], pairs[:, 0]))
        pairs, sw = np.ascontiguousarray(pairs[by]), np.ascontiguousarray(sw[by])
        perms = np.stack([order.epoch_positions(m.user_items.offsets, 100 + seed, e, seg_rows) for e in range(epochs)]).astype(np.int32)
    orc.fit(pairs, sw, m.user_items.offsets, m.user_items.items, m.x_uf, m.x_if, m.w_i, m.w_if, m.v_u, m.v_i, m.v_uf, m.v_if, m.alpha, m.beta,
            m.learning_rate, m.learning_schedule, m.learning_exponent, 1, epochs, perms=perms, rng_mode=orc.RNG_COUNTER, seed=100 + seed, membership="binary")
    return tag, seed, {k: getattr(m, k) for k in WEIGHTS}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--seeds", type=int, default=3)
    ap.add_argument("--factors", type=int, default=32)
    ap.add_argument("--epochs", type=int, default=5)
    ap.add_argument("--users", type=int, default=100_000)
    ap.add_argument("--items", type=int, default=50_000)
    ap.add_argument("--engine-variants", default="", help="';'-separated engine variants instead of the built-in list: 'hot_publications=96;damping=64,hot_publications=192'")
    ap.add_argument("--oracle-variants", default="0,32", help="segment lengths of the sequential oracle runs (0 = row-level shuffle)")
    a = ap.parse_args()
    from rankfm_amd import EngineOptions, RankFM, evaluation, synthetic
    t0 = time.time()
    data = {s: synthetic.make_planted_large_device(a.users, a.items, seed=s) for s in range(a.seeds)}
    print("data ready after %.1f s" % (time.time() - t0), flush=True)
    oracle_variants = {("oracle row shuffle" if int(r) == 0 else "oracle segments<=%d" % int(r)): int(r) for r in a.oracle_variants.split(",")}
    assert "oracle row shuffle" in oracle_variants
    pool = mp.get_context("spawn").Pool(len(oracle_variants) * a.seeds)
    pending = [pool.apply_async(oracle_job, ((tag, s, data[s]["train"], a.factors, a.epochs, rows),)) for tag, rows in oracle_variants.items() for s in data]
    engine_variants = {"engine default": {}, "engine segments<=16": dict(tune=dict(segment_rows=16)), "engine segments<=8": dict(tune=dict(segment_rows=8)),
                       "engine 128 workgroups": dict(n_workgroups=128), "engine 64 workgroups": dict(n_workgroups=64)}
    if a.engine_variants:
        engine_variants = {"engine default": {}}
        for text in a.engine_variants.split(";"):
            kw = dict(tune={})
            for part in text.split(","):
                k, v = part.split("=")
                if k == "workgroups":
                    kw["n_workgroups"] = int(v)
                elif k == "damping":
                    kw["damping"] = float(v)
                elif k == "flags":
                    kw["debug_flags"] = int(v)
                else:
                    kw["tune"][k] = int(v)
            engine_variants["engine " + text] = kw
    hits = {}
    frames = {s: (pd.DataFrame(d["train"], columns=["u", "i"]), pd.DataFrame(d["test"], columns=["u", "i"])) for s, d in data.items()}
    for tag, kw in engine_variants.items():
        for s in data:
            m = RankFM(factors=a.factors, loss="bpr", engine=EngineOptions(seed=100 + s, **kw))
            np.random.seed(s)
            m.fit(frames[s][0], epochs=a.epochs)
            hits.setdefault(tag, {})[s] = (evaluation.hit_rate(m, frames[s][1], k=10), float(np.linalg.norm(m.w_i)), float(np.mean(m.last_fit_report["sgd_kernel_ms"])))
    for p in pending:
        tag, s, w = p.get(timeout=3000)
        o = RankFM(factors=a.factors, loss="bpr", engine=EngineOptions(seed=100 + s))
        np.random.seed(s)
        o._init_all(frames[s][0])
        for k, v in w.items():
            setattr(o, k, np.ascontiguousarray(v))
        o.is_fit = True
        hits.setdefault(tag, {})[s] = (evaluation.hit_rate(o, frames[s][1], k=10), float(np.linalg.norm(o.w_i)), 0.0)
    pool.close()
    base = np.array([hits["oracle row shuffle"][s][0] for s in data])
    for tag, r in hits.items():
        h = np.array([r[s][0] for s in data])
        print("%-26s hit_rate@10 %s mean %.4f  vs the row-shuffled oracle %+.2f point  |w_i| %.2f  kernel ms %.3f"
              % (tag, np.round(h, 4).tolist(), h.mean(), 100 * (h.mean() - base.mean()), np.mean([r[s][1] for s in data]), np.mean([r[s][2] for s in data])), flush=True)
    print("total %.0f s" % (time.time() - t0))


if __name__ == "__main__":
    main()
