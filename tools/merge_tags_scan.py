"""GPU-box experiment: a model WITH TAGS (config 4's kind: 8 + 8 binary tags, the features kernels and the table trainer in every shard) trained as
`world` merged user shards (distributed.emulate_ranks_on_one_device) against ONE engine on the whole data: hit_rate@10 and the norms of all six
arrays.  Variants "world:syncs:tables[:late][:bf16]" separated by ';' -- tables = mean | one (emulate_ranks_on_one_device; the product merges the mean; the 'turns' rows of profiles/r06_notes.md section 8 are commit 84dc127's), bf16 = SharedTables.exchange_dtype.  No oracle (the one-GPU engine
is held to it by tests/test_gpu_quality.py); measurement tooling, not product.

    python tools/merge_tags_scan.py --users 100000 --variants "8:auto:one;8:auto:mean;8:2:one;8:1:one;2:auto:one" """
import argparse
import os
import sys
import time

import numpy as np
import pandas as pd

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
NAMES = ("v_u", "v_i", "w_i", "v_uf", "v_if", "w_if")


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--variants", default="8:auto:one;8:auto:mean;8:2:one;8:1:one;2:auto:one")
    ap.add_argument("--users", type=int, default=100_000)
    ap.add_argument("--items", type=int, default=50_000)
    ap.add_argument("--seeds", type=int, default=1)
    ap.add_argument("--factors", type=int, default=32)
    ap.add_argument("--epochs", type=int, default=10)
    ap.add_argument("--learning-rate", type=float, default=0.05)
    ap.add_argument("--no-tags", action="store_true", help="the same problem without the tags (no features kernels)")
    ap.add_argument("--tune", default="", help="rfm_fit_tuning fields of every shard's session: name=value,...")
    a = ap.parse_args()
    import torch
    from rankfm_amd import EngineOptions, RankFM, evaluation, synthetic
    from rankfm_amd.distributed import emulate_ranks_on_one_device
    tune = {k: int(v) for k, v in (kv.split("=") for kv in a.tune.split(",") if kv)}
    t0 = time.time()
    res = {}

    def score(tag, model):
        res.setdefault(tag, []).append([evaluation.hit_rate(model, test, k=10)] + [float(np.linalg.norm(getattr(model, k))) for k in NAMES])

    for s in range(a.seeds):
        d = synthetic.make_planted_large_device(a.users, a.items, seed=s, n_tags=8)
        train, test = pd.DataFrame(d["train"], columns=["u", "i"]), pd.DataFrame(d["test"], columns=["u", "i"])
        us, its = np.unique(d["train"][:, 0]), np.unique(d["train"][:, 1])
        uf = pd.concat([pd.DataFrame({"u": us}), pd.DataFrame(d["user_tags"][us])], axis=1)
        itf = pd.concat([pd.DataFrame({"i": its}), pd.DataFrame(d["item_tags"][its])], axis=1)
        if a.no_tags:
            uf = itf = None
        for es in (100 + s, 1100 + s):
            m = RankFM(factors=a.factors, loss="bpr", learning_rate=a.learning_rate, engine=EngineOptions(seed=es))
            np.random.seed(s)
            m.fit(train, uf, itf, epochs=a.epochs)
            score("one engine, whole data", m)
        print("seed %d: %d rows, one engine done at %.0f s" % (s, len(train), time.time() - t0), flush=True)
        for v in a.variants.split(";"):
            parts = v.split(":")
            world, syncs, tables, late = int(parts[0]), parts[1], parts[2], "late" in parts[3:]
            xd = "bf16" if "bf16" in parts[3:] else "fp32"
            m0 = RankFM(factors=a.factors, loss="bpr", learning_rate=a.learning_rate, engine=EngineOptions(seed=100 + s))
            np.random.seed(s)
            m0._init_all(train, uf, itf)
            problem = dict(interactions=m0.interactions, sample_weight=m0.sample_weight, csr_offsets=m0.user_items.offsets, csr_items=m0.user_items.items,
                           x_uf=m0.x_uf, x_if=m0.x_if, weights={k: np.array(getattr(m0, k), copy=True) for k in ("w_i", "w_if", "v_u", "v_i", "v_uf", "v_if")})
            hyper = dict(alpha=m0.alpha, beta=m0.beta, learning_rate=a.learning_rate, learning_schedule="constant", learning_exponent=0.25, max_samples=1)
            kw = dict(has_user_features=0 if a.no_tags else 1, has_item_features=0 if a.no_tags else 1)
            t = dict(tune)
            if "debug_flags" in t:
                kw["debug_flags"] = t.pop("debug_flags")
            if t:
                kw["tune"] = t
            out = emulate_ranks_on_one_device(problem, world, hyper, a.epochs, torch.device("cuda", 0), syncs_per_epoch=(syncs if syncs == "auto" else int(syncs)),
                                              seed=100 + s, late=late, table_merge=tables, exchange_dtype=xd, **kw)
            for k, w in out.items():
                setattr(m0, k, np.ascontiguousarray(w))
            m0.is_fit = True
            score(v, m0)
            print("   %s done at %.0f s" % (v, time.time() - t0), flush=True)
    base = np.mean(res["one engine, whole data"], axis=0)
    for tag, rows in res.items():
        r = np.mean(rows, axis=0)
        print("%-28s hit_rate@10 %.4f (%+.2f pt)  norms / one engine - 1: %s"
              % (tag, r[0], 100 * (r[0] - base[0]), "  ".join("%s %+.1f%%" % (n, 100 * (r[1 + q] / base[1 + q] - 1)) for q, n in enumerate(NAMES))), flush=True)
    print("total %.0f s" % (time.time() - t0))


if __name__ == "__main__":
    main()
