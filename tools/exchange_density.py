"""CPU-only count behind DESIGN.md section 8.1's "a smaller payload": how many item rows does ONE rank touch in ONE exchange window of
BASELINE config 4 (8 ranks)?  A lossless payload cut would exchange only those rows (index + row) instead of the dense bucket.  The
positives of a window are counted from the rank's shard in the engine's visiting order; a row's negative is uniform over the catalogue
(rankfm/_rankfm.pyx:250-253), so the expected share of items hit as a negative by n rows is 1 - exp(-n / I).

    python tools/exchange_density.py [--config C4] [--world 8] [--windows 8,24]

Measurement tooling, not product."""
import argparse
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from rankfm_amd import order, synthetic   # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--config", default="C4")
    ap.add_argument("--world", type=int, default=8)
    ap.add_argument("--windows", default="1,8,24")
    a = ap.parse_args()
    sh = synthetic.make_config_shard(a.config, rank=0, world=a.world)
    I = sh["x_if"].shape[0]
    n = len(sh["interactions"])
    pos = order.epoch_positions(sh["csr_offsets"], 1492, 0)
    items = sh["csr_items"][pos]
    F = sh["weights"]["v_i"].shape[1]
    dense = I * (F + 1) * 4
    print("%s, rank 0 of %d: %d rows, %d items, dense item-side payload %.1f MB" % (a.config, a.world, n, I, dense / 1e6))
    for w in [int(x) for x in a.windows.split(",")]:
        per = n // w
        shares = []
        for k in range(min(w, 4)):
            touched = np.zeros(I, bool)
            touched[items[k * per:(k + 1) * per]] = True
            p_pos = touched.mean()
            p_neg = 1.0 - np.exp(-per / I)
            shares.append(1.0 - (1.0 - p_pos) * (1.0 - p_neg))
        s = float(np.mean(shares))
        # a sparse exchange is an all-gather of (index, row) from every rank: world x touched rows x (F + 2) floats received per rank,
        # against ~2 x the dense bucket moved by a ring all-reduce
        sparse = a.world * s * I * (F + 2) * 4
        print("  %2d windows per epoch: %7d rows per window; rows touched (positive or negative): %.1f %% of the catalogue; all-gather of "
              "(index, row) from %d ranks: %.1f MB received per rank against ~%.1f MB moved by the dense ring all-reduce"
              % (w, per, 100 * s, a.world, sparse / 1e6, 2 * dense / 1e6))


if __name__ == "__main__":
    main()
