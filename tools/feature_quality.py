"""GPU-box experiment: the reference-backed feature-model quality problem of tests/test_gpu_quality.py (3000 x 2000 planted, 8 + 8 tags
that carry signal, k=20, lr 0.03, 5 epochs, five seeds) for a few engine variants, against the reference's own numbers
(tests/golden/quality_planted_tags.npz).   python tools/feature_quality.py "" "flags=4" "table_producers=2"
(test infrastructure)"""
import os
import sys

import numpy as np
import pandas as pd

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from rankfm_amd import EngineOptions, RankFM, evaluation, synthetic      # noqa: E402

z = np.load(os.path.join(ROOT, "tests", "golden", "quality_planted_tags.npz"))
ref = z["bpr"]
print("reference per seed hit", np.round(ref[:, 0], 4), "mean", np.round(ref[:, :7].mean(axis=0), 4), flush=True)
data = []
for seed in range(5):
    d = synthetic.make_planted(seed=seed, n_users=3000, n_items=2000, mean_degree=100.0, n_tags=8)
    data.append((pd.DataFrame(d["train"], columns=["u", "i"]), pd.DataFrame(d["test"], columns=["u", "i"]),
                 pd.DataFrame(np.column_stack([np.arange(len(d["user_tags"])), d["user_tags"]])),
                 pd.DataFrame(np.column_stack([np.arange(len(d["item_tags"])), d["item_tags"]]))))
RUNS = int(os.environ.get("FQ_RUNS", "1"))      # engine runs per data seed (one run's hit rate moves by +-2 points)
for variant in (sys.argv[1:] or [""]):
    kv = dict(x.split("=") for x in variant.split(",") if x)
    flags = int(kv.pop("flags", 0))
    got = []
    for seed, (train, test, uf, itf) in list(enumerate(data)) * RUNS:
        m = RankFM(factors=20, loss="bpr", learning_rate=0.03, engine=EngineOptions(debug_flags=flags, tune={k: int(v) for k, v in kv.items()}))
        np.random.seed(seed)
        m.fit(train, user_features=uf, item_features=itf, epochs=5)
        got.append([evaluation.hit_rate(m, test, k=10)] + [np.linalg.norm(getattr(m, k)) for k in ("v_u", "v_i", "w_i", "v_uf", "v_if", "w_if")])
    got = np.array(got)
    print("%-28s hit per run %s  mean %s  vs reference %s" % (variant or "default", np.round(got[:, 0], 3), np.round(got.mean(axis=0), 4),
                                                              np.round(got.mean(axis=0) / ref[:, :7].mean(axis=0) - 1.0, 4)), flush=True)
