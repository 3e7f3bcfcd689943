"""GPU-box measurement: BASELINE config 1's shape (MovieLens-1M-sized planted surrogate: 6,040 users x 3,706 items, ~1 M interactions,
k = 20, 5 epochs) through the public API -- SGD kernel ms per epoch and hit_rate@10, BPR and WARP.  python tools/c1_timing.py [runs]"""
import os
import sys

import numpy as np
import pandas as pd

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from rankfm_amd import RankFM, evaluation, synthetic      # noqa: E402

runs = int(sys.argv[1]) if len(sys.argv) > 1 else 2
d = synthetic.make_planted(seed=0)
train, test = pd.DataFrame(d["train"], columns=["u", "i"]), pd.DataFrame(d["test"], columns=["u", "i"])
deg = np.bincount(d["train"][:, 0])
print("train rows %d, users %d, mean degree %.0f, users with more than 64 / 128 / 256 items: %.0f %% / %.0f %% / %.0f %%"
      % (len(train), len(deg), deg.mean(), 100 * (deg > 64).mean(), 100 * (deg > 128).mean(), 100 * (deg > 256).mean()))
for loss in ("bpr", "warp"):
    for r in range(runs):
        m = RankFM(factors=20, loss=loss, max_samples=20)
        np.random.seed(0)
        m.fit(train, epochs=5)
        ms = m.last_fit_report["sgd_kernel_ms"]
        print("%-4s run %d: kernel ms per epoch %s  hit_rate@10 %.4f" % (loss, r, np.round(ms, 3).tolist(), evaluation.hit_rate(m, test, k=10)), flush=True)
