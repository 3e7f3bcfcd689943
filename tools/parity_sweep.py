"""GPU-box experiment: Hogwild engine vs the sequential CPU oracle on the same counter-based draws, for a sweep of
damping constants / concurrency.  Prints norm ratios, per-epoch LL ratios and throughput.  (uses oracle/: tooling, not product)"""
import argparse
import sys
import os
import time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import oracle as orc
from rankfm_amd import synthetic
from rankfm_amd.engine import DeviceSession

ap = argparse.ArgumentParser()
ap.add_argument("--users", type=int, default=100000)
ap.add_argument("--items", type=int, default=50000)
ap.add_argument("--rows", type=int, default=5000000)
ap.add_argument("--factors", type=int, default=64)
ap.add_argument("--epochs", type=int, default=2)
ap.add_argument("--zipf", type=float, default=1.0)
ap.add_argument("--max-samples", type=int, default=1)
ap.add_argument("--dampings", default="-1,16,64,256")
ap.add_argument("--workgroups", default="0")
ap.add_argument("--sigma", type=float, default=0.1)
a = ap.parse_args()
U, I, N, F = a.users, a.items, a.rows, a.factors
pairs, csr = synthetic.make_interactions(U, I, N, seed=0, zipf_s=a.zipf)
w = synthetic.init_weights(U, I, F, sigma=a.sigma, seed=1492)
sw = np.ones(N, np.float32)
x_uf, x_if = np.zeros((U, 1), np.float32), np.zeros((I, 1), np.float32)
o = {k: v.copy() for k, v in w.items()}
t0 = time.time()
out = orc.fit(pairs, sw, csr.offsets, csr.items, x_uf, x_if, o["w_i"], o["w_if"], o["v_u"], o["v_i"], o["v_uf"], o["v_if"],
              0.01, 0.1, 0.1, "constant", 0.25, a.max_samples, a.epochs, perms=None, rng_mode=orc.RNG_COUNTER, seed=1492,
              membership="binary")
print("oracle %.1fs ll/N %s  |w_i| %.3f |v_u| %.3f |v_i| %.3f" % (time.time() - t0, out["ll"] / N, np.linalg.norm(o["w_i"]),
      np.linalg.norm(o["v_u"]), np.linalg.norm(o["v_i"])), flush=True)
cnt = np.bincount(pairs[:, 1], minlength=I)
hot = np.argsort(-cnt)[:64]
for wg in [int(x) for x in a.workgroups.split(",")]:
    for m in [float(x) for x in a.dampings.split(",")]:
        sess = DeviceSession(pairs, sw, csr.offsets, csr.items, x_uf, x_if, w, max_samples=a.max_samples, seed=1492,
                             hogwild_damping=m, n_workgroups=wg)
        rep = sess.run(epochs=a.epochs)
        g = sess.weights_to_host()
        r = {k: np.linalg.norm(g[k]) / np.linalg.norm(o[k]) for k in ("w_i", "v_u", "v_i")}
        c = {k: np.corrcoef(g[k].ravel(), o[k].ravel())[0, 1] for k in ("w_i", "v_u", "v_i")}
        hot_ratio = np.linalg.norm(g["w_i"][hot]) / np.linalg.norm(o["w_i"][hot])
        print("wg=%4d M=%6.1f  ms %s  ll-ratio %s  norm w_i %.4f v_u %.4f v_i %.4f  corr %.4f %.4f %.4f  hot64 w_i ratio %.3f" % (
            wg, m, np.round(rep["sgd_kernel_ms"], 2), np.round(rep["log_likelihood"] / out["ll"], 4), r["w_i"], r["v_u"], r["v_i"],
            c["w_i"], c["v_u"], c["v_i"], hot_ratio), flush=True)
