"""CPU-only analysis: hit_rate@10 of the stripe kernel's execution model (oracle/rfm_async_sim.c) on the 30,000 x 12,000 planted
problem (k=20, 5 epochs), per seed and variant.
    [SEGROWS=16] python tools/model_quality.py 0,1,2 seq,rows=0 rows=0 w=24 seq,w=24 w=24,skew=12 w=24,ph=2,rows=94
`seq` = the sequential limit (negatives still follow the stripe schedule unless rows=0); SEGROWS = rows per user segment (32 in the
engine).  Numbers in profiles/r02_notes.md.  (test / analysis infrastructure: uses oracle/)"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import async_sim as sim
from rankfm_amd import synthetic, order
from rankfm_amd._rankfm import UserItemsCSR
U, I, F, E = 30000, 12000, 20, 5
order.SEGMENT_ROWS = int(os.environ.get('SEGROWS', '32'))
seeds = [int(x) for x in sys.argv[1].split(",")]
specs = sys.argv[2:]
for sd in seeds:
    d = synthetic.make_planted(U, I, seed=sd)
    pairs, test = d["train"], d["test"]
    N = len(pairs)
    csr = UserItemsCSR.from_pairs(pairs[:, 0], pairs[:, 1], U)
    tcsr = UserItemsCSR.from_pairs(test[:, 0], test[:, 1], U)
    by_csr = np.lexsort((pairs[:, 1], pairs[:, 0]))
    pairs_csr = np.ascontiguousarray(pairs[by_csr])
    sw = np.ones(N, np.float32)
    test_users = np.unique(test[:, 0])
    cnt = np.bincount(pairs[:, 1], minlength=I)
    n_seg = len(order.segments(csr.offsets)[0])
    def hit_rate(w, k=10):
        hits = 0
        for u0 in range(0, len(test_users), 1024):
            us = test_users[u0:u0 + 1024]
            S = w["v_u"][us] @ w["v_i"].T + w["w_i"]
            for r, u in enumerate(us):
                S[r, csr.items[csr.offsets[u]:csr.offsets[u + 1]]] = -np.inf
            top = np.argpartition(-S, k, axis=1)[:, :k]
            for r, u in enumerate(us):
                hits += bool(np.intersect1d(top[r], tcsr.items[tcsr.offsets[u]:tcsr.offsets[u + 1]]).size)
        return hits / len(test_users)
    for spec in specs:
        o = dict(x.split("=") for x in spec.split(",") if "=" in x)
        seq = "seq" in spec.split(",")
        geo = sim.default_geometry(U, I, N, n_seg, 64, stripes=int(o.get("rows", 1)) > 0, window_factor=8.0)
        if geo["stripe_rows"]:
            geo["stripe_window"] = int(o.get("w", geo["stripe_window"]))
            if "rows" in o: geo["stripe_rows"] = int(o["rows"])
        phases = int(o.get("ph", 1))
        w = synthetic.init_weights(U, I, F, seed=100 + sd)
        t0 = time.time()
        if seq:
            kw = dict(defer=False, mean_view=0.0, publish_now=True, phases=phases)
        else:
            pos, user, hot_slot, hot_period = sim.damping_plan(cnt, csr.offsets, geo, factors=F)
            skew = np.random.default_rng(1).integers(0, int(o["skew"]) + 1, geo["workgroups"]) if "skew" in o else None
            kw = dict(pos_step=pos, user_step=user, hot_slot=hot_slot, hot_period=hot_period, mean_view=float(o.get("mean", 1.0)), skew=skew, phases=phases)
        ll = [sim.epoch(pairs_csr, sw, csr.offsets, csr.items, w, 100 + sd, e, geo, **kw)[0] for e in range(E)]
        print("seed", sd, spec, (geo["workgroups"], geo["working_groups"], geo["stripe_rows"], geo["stripe_window"]), "hit_rate@10 %.4f" % hit_rate(w),
              "LL/N", np.round(np.array(ll) / N, 4)[[0, -1]], "norms", [round(float(np.linalg.norm(w[k])), 2) for k in ("v_u", "v_i", "w_i")], "%.0f s" % (time.time() - t0), flush=True)
