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
for epochs, flags in ((1, 0), (1, 0), (1, 0), (5, 0), (20, 0)):
    g = {k: v.copy() for k, v in w.items()}
    rep = {}
    t0 = time.perf_counter()
    _fit(pairs, sw, csr, x_uf, x_if, g["w_i"], g["w_if"], g["v_u"], g["v_i"], g["v_uf"], g["v_if"], 0.01, 0.1, 0.1, "constant", 0.25, 1,
         epochs, False, engine=EngineOptions(seed=1, debug_flags=flags), report=rep)
    dt = time.perf_counter() - t0
    print("host-buffer _fit: epochs=%d%s  wall %.1f ms  -> %.1f M updates/s (PCIe + plan inclusive); kernel ms/epoch %s" % (
        epochs, " (debug_flags %d)" % flags if flags else "", dt * 1e3, N * epochs / dt / 1e6, np.round(rep["sgd_kernel_ms"][:3], 2)), flush=True)

# the floor of the boundary: the same host buffers moved by plain copies (pageable numpy memory -> HBM and the weights back), nothing else
import torch
dev = torch.device("cuda", 0)
up = [torch.from_numpy(a) for a in (pairs, sw, csr.offsets, csr.items, w["v_u"], w["v_i"], w["w_i"])]
down_like = [w["v_u"], w["v_i"], w["w_i"]]
for _ in range(3):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    on = [t.to(dev) for t in up]
    torch.cuda.synchronize()
    t1 = time.perf_counter()
    back = [on[4].cpu(), on[5].cpu(), on[6].cpu()]
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    mb_up = sum(t.numel() * t.element_size() for t in up) / 1e6
    mb_dn = sum(a.nbytes for a in down_like) / 1e6
    print("plain copies of the same buffers: %.0f MB up %.2f ms (%.1f GB/s), %.0f MB down %.2f ms (%.1f GB/s)" % (
        mb_up, (t1 - t0) * 1e3, mb_up / (t1 - t0) / 1e3, mb_dn, (t2 - t1) * 1e3, mb_dn / (t2 - t1) / 1e3), flush=True)
