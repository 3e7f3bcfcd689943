"""GPU-box measurement: the PCIe-inclusive rate of the drop-in boundary (`_fit` on host numpy buffers: upload, plan, train, download)
next to the resident-in-HBM rate that bench.py reports.  BASELINE config 2."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from rankfm_amd import EngineOptions, synthetic
from rankfm_amd._rankfm import _fit
cfg = synthetic.CONFIGS["C2"]
U, I, N, F = cfg["n_users"], cfg["n_items"], cfg["n_interactions"], cfg["factors"]
pairs, csr = synthetic.make_interactions(U, I, N, seed=0)
w = synthetic.init_weights(U, I, F, seed=1492)
sw = np.ones(N, np.float32)
x_uf, x_if = np.zeros((U, 1), np.float32), np.zeros((I, 1), np.float32)
for epochs in (1, 1, 5, 20):
    g = {k: v.copy() for k, v in w.items()}
    rep = {}
    t0 = time.perf_counter()
    _fit(pairs, sw, csr, x_uf, x_if, g["w_i"], g["w_if"], g["v_u"], g["v_i"], g["v_uf"], g["v_if"], 0.01, 0.1, 0.1, "constant", 0.25, 1,
         epochs, False, engine=EngineOptions(seed=1), report=rep)
    dt = time.perf_counter() - t0
    print("host-buffer _fit: epochs=%d  wall %.1f ms  -> %.1f M updates/s (PCIe + plan inclusive); kernel ms/epoch %s" % (
        epochs, dt * 1e3, N * epochs / dt / 1e6, np.round(rep["sgd_kernel_ms"][:3], 2)), flush=True)
