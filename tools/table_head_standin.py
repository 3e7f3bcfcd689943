"""CPU experiment behind the opening launch of feature models (profiles/r03_notes.md section 7): the sequential oracle on config 4's
one-GPU share with its dense feature tables trained on every k-th visited row only -- `table_every` -- and a different k for the first
rows of the first epoch -- `table_head_every` / `table_head_rows` -- against the reference's every-row updates.  One epoch each, eight
variants in parallel, ~1-2 minutes.  Analysis tooling (uses oracle/), not product.

    python tools/table_head_standin.py
"""
import multiprocessing as mp
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import oracle as orc                      # noqa: E402
from rankfm_amd import synthetic                      # noqa: E402

WE = ("w_i", "w_if", "v_u", "v_i", "v_uf", "v_if")
SH = {}


def task(spec):
    name, kw = spec
    sh = SH["d"]
    lr = synthetic.CONFIGS["C4"]["learning_rate"]
    w = {k: np.array(sh["weights"][k], copy=True) for k in WE}
    t0 = time.time()
    out = orc.fit(sh["interactions"], sh["sample_weight"], sh["csr_offsets"], sh["csr_items"], sh["x_uf"], sh["x_if"], w["w_i"], w["w_if"],
                  w["v_u"], w["v_i"], w["v_uf"], w["v_if"], 0.01, 0.1, lr, "constant", 0.25, 1, 1, perms=None, rng_mode=orc.RNG_COUNTER, seed=1492,
                  membership="binary", **kw)
    return name, float(out["ll64"][0]), {k: float(np.linalg.norm(w[k])) for k in WE}, time.time() - t0


def main():
    orc.build()
    SH["d"] = synthetic.make_config_shard("C4", rank=0, world=8)
    N = len(SH["d"]["interactions"])
    specs = [("every row (reference)", {}), ("every 240th", dict(table_every=240)), ("every 80th", dict(table_every=80))]
    for frac, k in ((0.01, 8), (0.02, 15), (0.05, 15), (0.02, 30), (0.10, 40)):
        specs.append(("every %dth for the first %g %%, then every 240th" % (k, 100 * frac), dict(table_every=240, table_head_every=k, table_head_rows=int(frac * N))))
    with mp.get_context("fork").Pool(min(8, os.cpu_count() or 1)) as pool:
        res = dict((r[0], r) for r in pool.map(task, specs))
    ref = res[specs[0][0]]
    for name, _ in specs:
        _, ll, nm, dt = res[name]
        print("%-52s LL/ref - 1 %+6.2f%%   " % (name, 100 * (ll / ref[1] - 1)) + " ".join("%s %+6.2f%%" % (k, 100 * (nm[k] / ref[2][k] - 1)) for k in WE)
              + "   %.0f s" % dt, flush=True)


if __name__ == "__main__":
    main()
