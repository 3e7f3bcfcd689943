"""GPU-box experiment: Hogwild engine vs the sequential CPU oracle on the same order and counter-based draws, for a sweep of
damping constants / concurrency / windows.  Prints norm ratios, per-epoch LL ratios and throughput.
(uses oracle/: tooling, not product)"""
import argparse
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import oracle as orc
from rankfm_amd import order, synthetic
from rankfm_amd.engine import DeviceSession

ap = argparse.ArgumentParser()
ap.add_argument("--users", type=int, default=100000)
ap.add_argument("--items", type=int, default=50000)
ap.add_argument("--rows", type=int, default=5000000)
ap.add_argument("--factors", type=int, default=64)
ap.add_argument("--epochs", type=int, default=2)
ap.add_argument("--zipf", type=float, default=1.0)
ap.add_argument("--max-samples", type=int, default=1)
ap.add_argument("--uf", type=int, default=0)
ap.add_argument("--if", dest="itf", type=int, default=0)
ap.add_argument("--dampings", default="0")
ap.add_argument("--workgroups", default="0")
ap.add_argument("--rows-per-launch", default="0")
ap.add_argument("--sigma", type=float, default=0.1)
ap.add_argument("--no-oracle", action="store_true")
ap.add_argument("--debug-flags", default="0")
ap.add_argument("--lr", type=float, default=0.1)
ap.add_argument("--variants", default="", help="geometry-override variants (rfm_fit_tuning): 'segment_rows=16;hot_publications=24'")
a = ap.parse_args()
U, I, N, F = a.users, a.items, a.rows, a.factors
pairs, csr = synthetic.make_interactions(U, I, N, seed=0, zipf_s=a.zipf)
w = synthetic.init_weights(U, I, F, a.uf, a.itf, sigma=a.sigma, seed=1492)
sw = np.ones(N, np.float32)
x_uf = synthetic.make_features(U, a.uf, 2) if a.uf else np.zeros((U, 1), np.float32)
x_if = synthetic.make_features(I, a.itf, 3) if a.itf else np.zeros((I, 1), np.float32)
NAMES = ("w_i", "v_u", "v_i") + (("w_if", "v_uf", "v_if") if (a.uf or a.itf) else ())
o, out = None, None
if not a.no_oracle:
    by_csr = np.lexsort((pairs[:, 1], pairs[:, 0]))
    pairs_csr = np.ascontiguousarray(pairs[by_csr])
    perms = np.stack([order.epoch_positions(csr.offsets, 1492, e) for e in range(a.epochs)]).astype(np.int32)
    o = {k: v.copy() for k, v in w.items()}
    t0 = time.time()
    out = orc.fit(pairs_csr, sw, csr.offsets, csr.items, x_uf, x_if, o["w_i"], o["w_if"], o["v_u"], o["v_i"], o["v_uf"], o["v_if"],
                  0.01, 0.1, a.lr, "constant", 0.25, a.max_samples, a.epochs, perms=perms, rng_mode=orc.RNG_COUNTER, seed=1492,
                  membership="binary")
    print("oracle %.1fs ll/N %s  " % (time.time() - t0, out["ll"] / N) + " ".join("|%s| %.3f" % (k, np.linalg.norm(o[k])) for k in NAMES), flush=True)
for envs in a.variants.split(";"):
  tune = {x.split("=")[0]: int(x.split("=")[1]) for x in envs.split(",") if "=" in x}
  print("variant", tune, flush=True)
  for wg in [int(x) for x in a.workgroups.split(",")]:
    for rpl in [int(x) for x in a.rows_per_launch.split(",")]:
          for m, fl in [(float(x), int(y)) for x in a.dampings.split(",") for y in a.debug_flags.split(",")]:
              sess = DeviceSession(pairs, sw, csr.offsets, csr.items, x_uf, x_if, w, max_samples=a.max_samples, seed=1492, learning_rate=a.lr,
                                   hogwild_damping=m, n_workgroups=wg, rows_per_launch=rpl, debug_flags=fl, tune=tune)
              rep = sess.run(epochs=a.epochs, raise_on_error=False)
              g = sess.weights_to_host()
              line = "flags=%d " % fl + "wg=%4d rpl=%8d M=%6.1f st=%d launches=%d ms %s ll/N %s" % (wg, rpl, m, rep["status"], rep["launches_per_epoch"],
                                                                           np.round(rep["sgd_kernel_ms"], 2), np.round(rep["log_likelihood"] / N, 4))
              if o is not None:
                  line += "  ll-ratio %s  norm-ratio " % np.round(rep["log_likelihood"] / out["ll"], 4)
                  line += " ".join("%s %.4f" % (k, np.linalg.norm(g[k]) / max(np.linalg.norm(o[k]), 1e-30)) for k in NAMES)
                  line += "  corr " + " ".join("%.4f" % np.corrcoef(g[k].ravel(), o[k].ravel())[0, 1] for k in NAMES)
              else:
                  line += "  norms " + " ".join("%s %.4g" % (k, np.linalg.norm(g[k])) for k in NAMES)
              print(line, flush=True)
