"""Config 2 at full size: two epochs on the GPU against the sequential oracle on the same order and draws, for a few engine
settings (what each mechanism of the production kernel costs in log-likelihood / norms).  Test infrastructure (uses oracle/).
    python tools/c2_ll_ratio.py [flags:damping[:window] ...]     e.g.  0:0 8:0 0:-1 0:0:8
(debug_flags bit 3 = no stripes; damping -1 = off; window = rows per stripe window, the RFM_STRIPE_WINDOW experiment knob)
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

from oracle import oracle as orc                      # noqa: E402
from rankfm_amd import synthetic                      # noqa: E402
from rankfm_amd.engine import DeviceSession           # noqa: E402
from test_gpu_parity import _oracle_in_engine_order   # noqa: E402

orc.build()
cfg = synthetic.CONFIGS["C2"]
U, I, N, F = cfg["n_users"], cfg["n_items"], cfg["n_interactions"], cfg["factors"]
pairs, csr = synthetic.make_interactions(U, I, N, seed=0)
sw = np.ones(N, np.float32)
x_uf, x_if = np.zeros((U, 1), np.float32), np.zeros((I, 1), np.float32)
cache = {}
for spec in (sys.argv[1:] or ["0:0"]):
    flags, damping, window = (spec.split(":") + [""])[:3]
    os.environ.pop("RFM_STRIPE_WINDOW", None)
    if window:
        os.environ["RFM_STRIPE_WINDOW"] = window
    w = synthetic.init_weights(U, I, F, seed=1492)
    sess = DeviceSession(pairs, sw, csr.offsets, csr.items, x_uf, x_if, w, max_samples=1, seed=1492, debug_flags=int(flags),
                         hogwild_damping=float(damping))
    rep = sess.run(epochs=2)
    g = sess.weights_to_host()
    geo = sess.geometry()
    key = (geo.get("stripe_rows"), geo.get("stripe_window"), geo.get("workgroups"))
    if key not in cache:
        cache[key] = _oracle_in_engine_order(orc, (pairs, csr, sw, x_uf, x_if, None), w, 1, 2, 1492, geometry=geo)
    o, out = cache[key]
    print(spec, "stripes", key, "LL gpu/oracle - 1 =", np.round(rep["log_likelihood"] / out["ll"] - 1.0, 5),
          "norms - 1 =", [round(float(np.linalg.norm(g[k]) / np.linalg.norm(o[k]) - 1.0), 5) for k in ("v_u", "v_i", "w_i")],
          "kernel ms", np.round(rep.get("sgd_kernel_ms", [0]), 3), flush=True)
