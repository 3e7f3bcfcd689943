"""GPU-box measurement: the full-size config-2 parity figures (log-likelihood of epochs 1 / 2 and norms against the sequential oracle on
the engine's order, tests/test_gpu_parity.py::test_hogwild_full_size_config2_tracks_sequential_oracle) as a function of the hot rows'
publications per epoch and workgroup.  Test infrastructure (uses oracle/), not product.

    python tools/pub_margin.py [--pubs 48,32,24] [--runs 2]"""
import argparse
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--pubs", default="48,32,24")
    ap.add_argument("--tune", default="", help="further overrides applied to every run: 'hot_sweep_every=4'")
    ap.add_argument("--runs", type=int, default=2)
    ap.add_argument("--flags", type=int, default=0)
    a = ap.parse_args()
    from oracle import oracle
    from rankfm_amd import synthetic
    from rankfm_amd.engine import DeviceSession
    from test_gpu_parity import _oracle_in_engine_order
    cfg = synthetic.CONFIGS["C2"]
    U, I, N, F = cfg["n_users"], cfg["n_items"], cfg["n_interactions"], cfg["factors"]
    pairs, csr = synthetic.make_interactions(U, I, N, seed=0)
    w = synthetic.init_weights(U, I, F, seed=1492)
    sw = np.ones(N, np.float32)
    x_uf, x_if = np.zeros((U, 1), np.float32), np.zeros((I, 1), np.float32)
    prob = (pairs, csr, sw, x_uf, x_if, None)
    oo = oout = None
    for pub in [int(p) for p in a.pubs.split(",")]:
        for run in range(a.runs):
            tune = {"hot_publications": pub}
            tune.update({kv.split("=")[0]: int(kv.split("=")[1]) for kv in a.tune.split(",") if kv})
            sess = DeviceSession(pairs, sw, csr.offsets, csr.items, x_uf, x_if, w, max_samples=1, seed=1492, tune=tune,
                                 debug_flags=a.flags)
            rep = sess.run(epochs=2)
            g = sess.weights_to_host()
            if oo is None:
                oo, oout = _oracle_in_engine_order(oracle, prob, w, 1, 2, 1492, geometry=sess.geometry())
            print("publications %3d run %d: LL gpu/oracle - 1 = %+.4f %% / %+.4f %%   norms v_u %+.3f %% v_i %+.3f %% w_i %+.3f %%   kernel ms %.3f / %.3f"
                  % (pub, run, 100 * (rep["log_likelihood"][0] / oout["ll64"][0] - 1), 100 * (rep["log_likelihood"][1] / oout["ll64"][1] - 1),
                     *[100 * (np.linalg.norm(g[k]) / np.linalg.norm(oo[k]) - 1) for k in ("v_u", "v_i", "w_i")], *rep["sgd_kernel_ms"]), flush=True)
            del sess


if __name__ == "__main__":
    main()
