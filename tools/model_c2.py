"""CPU-only analysis: the execution model of the stripe kernel (oracle/rfm_async_sim.c) on BASELINE config 2 at full size, two epochs,
against the sequential limit of the same schedule (same draws, everything visible at once, log-likelihood summed in double).
    [SEGROWS=16] python tools/model_c2.py w=24 w=24,skew=12 w=24,rows=97,ph=2,skew=24 rows=0 ...
keys: w = stripe window (rows per group), rows = stripe rows (0: no stripes), skew = workgroups lag by up to that many rounds,
ph = phases of the stripe schedule (experiment), mean = factor of the mean-field view of the positive item.
Numbers in profiles/r02_notes.md.  (test / analysis infrastructure: uses oracle/)"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import async_sim as sim
from rankfm_amd import synthetic, order
cfg = synthetic.CONFIGS["C2"]
U, I, N, F = cfg["n_users"], cfg["n_items"], cfg["n_interactions"], cfg["factors"]
pairs, csr = synthetic.make_interactions(U, I, N, seed=0)
sw = np.ones(N, np.float32)
by_csr = np.lexsort((pairs[:, 1], pairs[:, 0]))
pairs_csr = np.ascontiguousarray(pairs[by_csr])
E, seed = 2, 1492
order.SEGMENT_ROWS = int(os.environ.get("SEGROWS", "32"))      # rows per user segment (32 in the engine)
n_seg = len(order.segments(csr.offsets)[0])
cnt = np.bincount(pairs[:, 1], minlength=I)
def run(geo, E, **kw):
    w = synthetic.init_weights(U, I, F, seed=1492)
    ll = [sim.epoch(pairs_csr, sw, csr.offsets, csr.items, w, seed, e, geo, **kw)[0] for e in range(E)]
    return np.array(ll), w
base = {}
for spec in sys.argv[1:]:
    o = dict(x.split("=") for x in spec.split(",") if "=" in x)
    geo = sim.default_geometry(U, I, N, n_seg, F, stripes=int(o.get("rows", 194)) > 0, window_factor=8.0)
    if geo["stripe_rows"]:
        geo["stripe_window"] = int(o.get("w", 24)); geo["stripe_rows"] = int(o.get("rows", 194))
    phases = int(o.get("ph", 1))
    key = (geo["stripe_rows"], geo["stripe_window"], phases)
    t0 = time.time()
    if key not in base:
        base[key] = run(geo, E, defer=False, mean_view=0.0, phases=phases, publish_now=True)          # the sequential limit on the same draws
    pos, user, hot_slot, hot_period = sim.damping_plan(cnt, csr.offsets, geo)
    skew = np.random.default_rng(1).integers(0, int(o["skew"]) + 1, geo["workgroups"]) if "skew" in o else None
    ll, w = run(geo, E, pos_step=pos, user_step=user, hot_slot=hot_slot, hot_period=hot_period, mean_view=float(o.get("mean", 1.0)), skew=skew, phases=phases)
    nr = [float(np.linalg.norm(w[k]) / np.linalg.norm(base[key][1][k]) - 1) for k in ("v_u", "v_i", "w_i")]
    print(spec, key, "LL / sequential - 1", np.round(ll / base[key][0] - 1, 4), "norms", np.round(nr, 4), "%.0f s" % (time.time() - t0), flush=True)
