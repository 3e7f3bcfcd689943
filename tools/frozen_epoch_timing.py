"""GPU-box experiment: the SGD launch time of ONE epoch that starts from saved (trained) weights -- for timing-only builds of a kernel
whose results are wrong on purpose (e.g. a build that leaves the row updates out): such a build cannot train its own way to the
draws-per-row distribution the real kernel sees, so it is handed the weights a real build trained.

    python tools/frozen_epoch_timing.py --config C3 --train 6 --save /tmp/w.npz      # with the real library
    python tools/frozen_epoch_timing.py --config C3 --train 6 --load /tmp/w.npz      # with the build under test (also the real one: its A side)

Measurement tooling, not product."""
import argparse
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from rankfm_amd import synthetic                       # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--config", default="C3")
    ap.add_argument("--train", type=int, default=6)
    ap.add_argument("--save")
    ap.add_argument("--load")
    ap.add_argument("--reps", type=int, default=4)
    a = ap.parse_args()
    cfg = synthetic.CONFIGS[a.config]
    if a.config in ("C4", "C5"):
        sh = synthetic.make_config_shard(a.config, rank=0, world=8, zipf_s=1.0)
        data = (sh["interactions"], sh["sample_weight"], sh["csr_offsets"], sh["csr_items"], sh["x_uf"], sh["x_if"])
        weights = sh["weights"]
        n_uf, n_if = cfg.get("n_user_features", 0), cfg.get("n_item_features", 0)
    else:
        U, I, N, F = cfg["n_users"], cfg["n_items"], cfg["n_interactions"], cfg["factors"]
        pairs, csr = synthetic.make_interactions(U, I, N, seed=0, zipf_s=1.0)
        data = (pairs, np.ones(N, np.float32), csr.offsets, csr.items, np.zeros((U, 1), np.float32), np.zeros((I, 1), np.float32))
        weights = synthetic.init_weights(U, I, F, seed=1492)
        n_uf = n_if = 0
    N = len(data[0])
    from rankfm_amd.engine import DeviceSession
    hyper = dict(learning_rate=cfg.get("learning_rate", 0.1), max_samples=cfg["max_samples"], seed=1492,
                 has_user_features=int(n_uf > 0), has_item_features=int(n_if > 0))
    if a.save:
        s = DeviceSession(*data, {k: np.array(v, copy=True) for k, v in weights.items()}, **hyper)
        rep = s.run(epochs=a.train)
        np.savez(a.save, **{k: v.detach().cpu().numpy() for k, v in s.weights.items()})
        print("trained %d epochs: kernel ms %s draws/N %.2f" % (a.train, " ".join("%.3f" % x for x in rep["sgd_kernel_ms"]), rep["n_draws"][-1] / N))
        return
    saved = dict(np.load(a.load))
    for r in range(a.reps):
        s = DeviceSession(*data, {k: np.array(saved[k], copy=True) for k in weights}, **hyper)
        rep = s.run(epochs=1, epoch_begin=a.train)
        print("epoch %d from the saved weights: kernel ms %.3f  draws/N %.2f  LL/N %.5f"
              % (a.train, rep["sgd_kernel_ms"][0], rep["n_draws"][0] / N, rep["log_likelihood"][0] / N), flush=True)


if __name__ == "__main__":
    main()
