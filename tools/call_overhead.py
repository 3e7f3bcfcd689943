"""GPU-box measurement: what a resident `rfm_fit_device` call costs beyond its SGD launches (config 2).  Wall time of DeviceSession.run for
several epoch counts K, best of a few repetitions each; wall(K) = per-call overhead + K x (SGD launch + per-epoch overhead), the SGD
launches' own time from the call's HIP events.  Measurement tooling, not product.

    python tools/call_overhead.py [--epochs 1,2,5,10,20,40] [--reps 5] [--no-keep]"""
import argparse
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch                                           # noqa: E402
from rankfm_amd import synthetic                       # noqa: E402
from rankfm_amd.engine import DeviceSession            # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--epochs", default="1,2,5,10,20,40")
    ap.add_argument("--reps", type=int, default=5)
    ap.add_argument("--no-keep", action="store_true")
    ap.add_argument("--no-check", action="store_true")
    a = ap.parse_args()
    cfg = synthetic.CONFIGS["C2"]
    U, I, N, F = cfg["n_users"], cfg["n_items"], cfg["n_interactions"], cfg["factors"]
    pairs, csr = synthetic.make_interactions(U, I, N, seed=0)
    w = synthetic.init_weights(U, I, F, seed=1492)
    s = DeviceSession(pairs, np.ones(N, np.float32), csr.offsets, csr.items, np.zeros((U, 1), np.float32), np.zeros((I, 1), np.float32), w,
                      seed=1492, keep_layout=not a.no_keep, check_finite=not a.no_check)
    ks = [int(x) for x in a.epochs.split(",")]
    s.run(epochs=max(ks))                              # (workspace sized once, plan built)
    s.run(epochs=2)
    rows = []
    e = 100
    for k in ks:
        best = None
        for _ in range(a.reps):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            rep = s.run(epochs=k, epoch_begin=e)
            torch.cuda.synchronize()
            wall = (time.perf_counter() - t0) * 1e3
            kern = float(np.sum(rep["sgd_kernel_ms"]))
            if best is None or wall < best[0]:
                best = (wall, kern)
            e += k
        rows.append((k, best[0], best[1]))
        print("K = %3d: wall %8.3f ms, SGD launches %8.3f ms, wall - launches %7.3f ms = %6.1f us per epoch" % (
            k, best[0], best[1], best[0] - best[1], (best[0] - best[1]) / k * 1e3), flush=True)
    k = np.array([r[0] for r in rows], float)
    over = np.array([r[1] - r[2] for r in rows])
    b, c = np.polyfit(k, over, 1)
    print("fit: wall - launches = %.3f ms per call + %.1f us per epoch" % (c, b * 1e3))


if __name__ == "__main__":
    main()
