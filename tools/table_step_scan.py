"""GPU-box measurement for DESIGN.md 3.3 / VERDICT r04 item 2: how the dense feature tables of config 4's share respond to the table
trainer's STEP LENGTH (tune_table_step_pct) and QUOTA (tune_table_every) -- norms of v_uf / v_if / w_if after the second epoch against the
sequential oracle from the same start (the figures tests/test_gpu_configs.py holds to a tolerance).  Test infrastructure (uses oracle/).

    python tools/table_step_scan.py [--steps 100,75,50] [--every 0,223,892]"""
import argparse
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", default="100,75,50")
    ap.add_argument("--every", default="0")
    a = ap.parse_args()
    from oracle import oracle
    from rankfm_amd import synthetic
    from rankfm_amd.engine import DeviceSession
    from test_gpu_configs import _norm_ratio, _oracle_epoch
    oracle.build()
    sh = synthetic.make_config_shard("C4", rank=0, world=8)
    lr = sh["config"]["learning_rate"]
    ref = None
    for every in [int(x) for x in a.every.split(",")]:
        for pct in [int(x) for x in a.steps.split(",")]:
            tune = {"table_step_pct": pct}
            if every:
                tune["table_every"] = every
            sess = DeviceSession(sh["interactions"], sh["sample_weight"], sh["csr_offsets"], sh["csr_items"], sh["x_uf"], sh["x_if"],
                                 sh["weights"], max_samples=1, seed=1492, learning_rate=lr, tune=tune)
            rep1 = sess.run(epochs=1)
            g1 = sess.weights_to_host()
            rep2 = sess.run(epochs=1, epoch_begin=1)
            g2 = sess.weights_to_host()
            geo = sess.geometry()
            # the oracle's second epoch from THIS run's weights after the first (the tables' memory is a fraction of an epoch: what is
            # compared is the state the tables are left in, not a trajectory)
            o2 = {k: v.copy() for k, v in g1.items()}
            out2 = _oracle_epoch(oracle, sh, o2, 1, 1, 1492, lr, geo)
            r2 = {k: _norm_ratio(g2[k], o2[k]) for k in ("v_uf", "v_if", "w_if", "w_i", "v_u", "v_i")}
            print("step %3d %% every %4s (%d steps applied, tables kernel %s us): epoch-2 LL %+.3f %%  v_uf %.3f v_if %.3f w_if %.3f | w_i %.4f v_u %.4f v_i %.4f  kernel %.2f ms"
                  % (pct, every or "auto", geo["table_steps"], geo.get("table_span_us"), 100 * (rep2["log_likelihood"][0] / out2["ll64"][0] - 1),
                     r2["v_uf"], r2["v_if"], r2["w_if"], r2["w_i"], r2["v_u"], r2["v_i"], rep2["sgd_kernel_ms"][0]), flush=True)
            del sess


if __name__ == "__main__":
    main()
