"""GPU-box probe: the tags variant of tests/test_gpu_quality.py (config-2 shape, k = 32, 8 + 8 tags, lr 0.05, 10 epochs), several engine
runs per data seed and per VARIANT of the table trainer's overrides (rfm_fit_tuning) -- how much of the engine's hit_rate@10 is
run-to-run noise, and what a setting changes.  The data of a seed are generated once for all variants.

    python tools/tags_runs.py <runs> "<variant>;<variant>;..." [users:100000 items:50000 tags:8 factors:32 lr:0.05 epochs:10 seeds:3]

A variant is a ','-separated list of  <tune name>=<int>  (empty = the defaults); `every_x=<float>` scales the default quota's spacing
(it is resolved per seed from a default run's report).  Measurement tooling, not product."""
import os
import sys

import numpy as np
import pandas as pd

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from rankfm_amd import EngineOptions, RankFM, evaluation, synthetic   # noqa: E402

runs = int(sys.argv[1]) if len(sys.argv) > 1 else 3
variants = sys.argv[2].split(";") if len(sys.argv) > 2 else [""]
opts = dict(users=100_000, items=50_000, tags=8, factors=32, lr=0.05, epochs=10, seeds=3)
for x in sys.argv[3:]:
    k, v = x.split(":")
    opts[k] = type(opts[k])(v)
summary = {v: [] for v in variants}
for seed in range(opts["seeds"]):
    d = synthetic.make_planted_large_device(opts["users"], opts["items"], seed=seed, n_tags=opts["tags"])
    train, test = pd.DataFrame(d["train"], columns=["u", "i"]), pd.DataFrame(d["test"], columns=["u", "i"])
    us, its = np.unique(d["train"][:, 0]), np.unique(d["train"][:, 1])
    uf = pd.concat([pd.DataFrame({"u": us}), pd.DataFrame(d["user_tags"][us])], axis=1)
    itf = pd.concat([pd.DataFrame({"i": its}), pd.DataFrame(d["item_tags"][its])], axis=1)
    every_default = None
    for variant in variants:
        tune, every_x = {}, None
        for part in [p for p in variant.split(",") if p]:
            k, v = part.split("=")
            if k == "every_x":
                every_x = float(v)
            else:
                tune[k] = int(v)
        out = []
        for r in range(runs):
            if every_x is not None:
                if every_default is None:
                    probe = RankFM(factors=opts["factors"], loss="bpr", learning_rate=opts["lr"], engine=EngineOptions(seed=100 + seed))
                    np.random.seed(seed)
                    probe.fit(train, uf, itf, epochs=opts["epochs"])
                    every_default = len(train) * opts["epochs"] / max(probe.last_fit_report["geometry"]["table_steps"], 1)
                tune["table_every"] = max(1, int(round(every_default * every_x)))
            m = RankFM(factors=opts["factors"], loss="bpr", learning_rate=opts["lr"], engine=EngineOptions(seed=100 + seed + 1000 * r, tune=tune))
            np.random.seed(seed)
            m.fit(train, uf, itf, epochs=opts["epochs"])
            g = m.last_fit_report["geometry"]
            out.append((evaluation.hit_rate(m, test, k=10), float(np.linalg.norm(m.w_i)), float(np.linalg.norm(m.v_uf)), float(np.linalg.norm(m.v_if)),
                        float(np.linalg.norm(m.w_if)), float(np.mean(m.last_fit_report["sgd_kernel_ms"])), g.get("table_steps", 0), g.get("table_span_us", [0, 0])))
        summary[variant] += [o[0] for o in out]
        print("seed %d [%s] %s: hit_rate@10 %s mean %.4f | |w_i| %.2f |v_uf| %.3f |v_if| %.3f |w_if| %.3f | kernel ms %.3f, one table step per %.0f rows, tables / rows kernel us %s"
              % (seed, variant or "default", tune, [round(o[0], 4) for o in out], np.mean([o[0] for o in out]), np.mean([o[1] for o in out]),
                 np.mean([o[2] for o in out]), np.mean([o[3] for o in out]), np.mean([o[4] for o in out]), np.mean([o[5] for o in out]),
                 len(train) * opts["epochs"] / max(out[0][6], 1), out[0][7]), flush=True)
print("---- means over %d seeds x %d runs" % (opts["seeds"], runs))
for v in variants:
    print("%-40s hit_rate@10 %.4f  (sigma of a run %.4f)" % (v or "default", np.mean(summary[v]), np.std(summary[v], ddof=1)))
