"""Ranking-quality golden numbers from the REFERENCE itself (build container only; see make_golden.py for how the reference is
loaded).  BASELINE.json config 1 hyper-parameters (factors=20, loss='bpr', epochs=5) on the seeded planted MovieLens-1M-shaped
surrogate (rankfm_amd.synthetic.make_planted; real ML-1M is not available offline): per seed the reference's own
evaluation.hit_rate / precision / recall @10 and the Frobenius norms of its fitted factors.  Also WARP (max_samples=20).
Only numbers are stored; the data are regenerated from the seeds on the GPU box."""
import os
import sys
import time

import numpy as np
import pandas as pd

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.join(HERE, "..", "..")
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))
import ref_loader  # noqa: E402
from rankfm_amd import synthetic  # noqa: E402

RankFM, _, ev = ref_loader.load_reference()
out = {}
for loss in ("bpr", "warp"):
    rows = []
    for seed in range(5):
        d = synthetic.make_planted(seed=seed)
        train, test = pd.DataFrame(d["train"], columns=["u", "i"]), pd.DataFrame(d["test"], columns=["u", "i"])
        m = RankFM(factors=20, loss=loss, max_samples=20)
        np.random.seed(seed)
        t0 = time.time()
        m.fit(train, epochs=5)
        t_fit = time.time() - t0
        rows.append([ev.hit_rate(m, test, k=10), ev.precision(m, test, k=10), ev.recall(m, test, k=10),
                     np.linalg.norm(m.v_u), np.linalg.norm(m.v_i), np.linalg.norm(m.w_i), t_fit, len(train)])
        print(loss, seed, np.round(rows[-1], 4), flush=True)
    out[loss] = np.array(rows)
np.savez(os.path.join(HERE, "quality_planted.npz"), columns=np.array(["hit_rate", "precision", "recall", "norm_v_u", "norm_v_i", "norm_w_i",
                                                                        "fit_seconds", "n_train"]), **out)
