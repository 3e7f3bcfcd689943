"""GPU-box experiment: does a SCHEDULE of samplers keep the ranking-quality bar -- the opt-in negative stripes (1.5 x faster) for part of
the epochs and the reference's uniform sampler for the rest?  Planted surrogate of make_planted_large, hit_rate@10 on held-out pairs
against the sequential oracle with the reference's sampler (tools/quality_parity.py).  A schedule is a string of 'u' / 's' per epoch.
    python tools/sampler_schedule.py --users 30000 --items 12000 --seeds 5 --schedules uuuuuuuu,ssssssss,uussssss,ssssssuu,ssssuuuu,susususu
(uses oracle/: tooling, not product)"""
import argparse
import os
import sys
import time

import numpy as np
import pandas as pd

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from _gpulock import gpu                                   # noqa: E402
from oracle import oracle as orc                           # noqa: E402
from rankfm_amd import EngineOptions, RankFM, evaluation, synthetic    # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--users", type=int, default=30000)
ap.add_argument("--items", type=int, default=12000)
ap.add_argument("--factors", type=int, default=32)
ap.add_argument("--seeds", type=int, default=5)
ap.add_argument("--degree", type=float, default=60.0)
ap.add_argument("--schedules", default="uuuuuuuu,ssssssss,uussssss,ssssssuu,ssssuuuu,susususu")
ap.add_argument("--no-oracle", action="store_true")
a = ap.parse_args()
scheds = a.schedules.split(",")
E = len(scheds[0])
res = {s: [] for s in scheds + ["oracle"]}
T0 = time.time()
for seed in range(a.seeds):
    d = synthetic.make_planted_large(a.users, a.items, seed=seed, mean_degree=a.degree)
    train, test = pd.DataFrame(d["train"], columns=["u", "i"]), pd.DataFrame(d["test"], columns=["u", "i"])
    if not a.no_oracle:
        m = RankFM(factors=a.factors, loss="bpr", engine=EngineOptions(seed=100 + seed))
        np.random.seed(seed)
        m._init_all(train, None, None, None)
        orc.fit(m.interactions, m.sample_weight, m.user_items.offsets, m.user_items.items, m.x_uf, m.x_if, m.w_i, m.w_if, m.v_u, m.v_i, m.v_uf, m.v_if,
                m.alpha, m.beta, m.learning_rate, m.learning_schedule, m.learning_exponent, 1, E, perms=None, rng_mode=orc.RNG_COUNTER, seed=100 + seed,
                membership="binary")
        m.is_fit = True
        res["oracle"].append(evaluation.hit_rate(m, test, k=10))
    for s in scheds:
        m = RankFM(factors=a.factors, loss="bpr", engine=EngineOptions(seed=100 + seed))
        np.random.seed(seed)
        k = 0
        with gpu():
            while k < E:                                   # runs of equal letters = one fit / fit_partial call each
                n = 1
                while k + n < E and s[k + n] == s[k]:
                    n += 1
                m.engine = EngineOptions(seed=100 + seed, negative_stripes=s[k] == "s")
                (m.fit if k == 0 else m.fit_partial)(train, epochs=n)
                k += n
        res[s].append(evaluation.hit_rate(m, test, k=10))
    print("seed %d done after %.0f s: " % (seed, time.time() - T0) + "  ".join("%s %.4f" % (k, v[-1]) for k, v in res.items() if v), flush=True)
o = np.mean(res["oracle"]) if res["oracle"] else float("nan")
for s in scheds:
    print("%-10s hit_rate@10 %.4f +- %.4f   vs oracle %.4f: %+.2f point" % (s, np.mean(res[s]), np.std(res[s]), o, 100 * (np.mean(res[s]) - o)))
