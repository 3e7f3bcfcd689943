"""Config 2 at full size: two epochs on the GPU against the sequential oracle on the same order and draws, for a few engine
settings (what each mechanism of the production kernel costs in log-likelihood / norms).  Test infrastructure (uses oracle/).
    python tools/c2_ll_ratio.py [flags:damping[:window] ...]     e.g.  0:0 8:0 0:-1 0:0:8
(debug_flags bit 3 = no stripes; damping -1 = off; window = rows per stripe window, the RFM_STRIPE_WINDOW experiment knob)
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

from oracle import oracle as orc                      # noqa: E402
from rankfm_amd import synthetic                      # noqa: E402
from rankfm_amd.engine import DeviceSession           # noqa: E402
from test_gpu_parity import _oracle_in_engine_order   # noqa: E402

orc.build()
cfg = synthetic.CONFIGS["C2"]
U, I, N, F = cfg["n_users"], cfg["n_items"], cfg["n_interactions"], cfg["factors"]
pairs, csr = synthetic.make_interactions(U, I, N, seed=0)
sw = np.ones(N, np.float32)
x_uf, x_if = np.zeros((U, 1), np.float32), np.zeros((I, 1), np.float32)


def step_scales(sess, damping):
    """the Hogwild step damping of the session's last run in the form the damped oracle takes (oracle.fit pos_step / user_step):
    the per-item scale of the positive item's step decoded from the plan at the head of the engine's workspace (rfm_api.hip `carve`:
    pos_scale [I] comes first; hot slots are encoded as scale + 2 (slot + 1)) and the per-user scale min(1, user_cap / degree) with
    user_cap = M x segments / interactions in flight (rfm_api.hip "plan, part 3").  NOT yet run on hardware (written after the
    round's GPU budget was spent)."""
    import torch
    g = sess.geometry()
    m = 128.0 if damping == 0 else damping
    if m <= 0 or g.get("single_group"):
        return np.ones(I, np.float32), np.ones(U, np.float32)
    raw = sess._workspace[:4 * I].view(torch.float32).cpu().numpy().astype(np.float64)
    slot = np.where(raw >= 2.0, np.floor(raw * 0.5), 0.0)
    pos = (raw - 2.0 * slot).astype(np.float32)
    user_cap = m * float(g["n_units"]) / float(g["working_groups"])
    deg = np.maximum(np.diff(csr.offsets), 1)
    return pos, np.minimum(1.0, user_cap / deg).astype(np.float32)


cache = {}
DAMPED = bool(os.environ.get("RFM_DAMPED_ORACLE"))      # also compare with the sequential oracle under the same step damping
for spec in (sys.argv[1:] or ["0:0"]):
    flags, damping, window = (spec.split(":") + [""])[:3]
    os.environ.pop("RFM_STRIPE_WINDOW", None)
    if window:
        os.environ["RFM_STRIPE_WINDOW"] = window
    w = synthetic.init_weights(U, I, F, seed=1492)
    sess = DeviceSession(pairs, sw, csr.offsets, csr.items, x_uf, x_if, w, max_samples=1, seed=1492, debug_flags=int(flags),
                         hogwild_damping=float(damping))
    rep = sess.run(epochs=2)
    g = sess.weights_to_host()
    geo = sess.geometry()
    key = (geo.get("stripe_rows"), geo.get("stripe_window"), geo.get("workgroups"))
    if key not in cache:
        cache[key] = _oracle_in_engine_order(orc, (pairs, csr, sw, x_uf, x_if, None), w, 1, 2, 1492, geometry=geo)
    o, out = cache[key]
    print(spec, "stripes", key, "LL gpu/oracle - 1 =", np.round(rep["log_likelihood"] / out["ll"] - 1.0, 5),
          "norms - 1 =", [round(float(np.linalg.norm(g[k]) / np.linalg.norm(o[k]) - 1.0), 5) for k in ("v_u", "v_i", "w_i")],
          "kernel ms", np.round(rep.get("sgd_kernel_ms", [0]), 3), flush=True)
    if DAMPED:
        from rankfm_amd import order
        pos_step, user_step = step_scales(sess, float(damping))
        by_csr = np.lexsort((pairs[:, 1], pairs[:, 0]))
        perms = np.stack([order.epoch_positions(csr.offsets, 1492, e, geo.get("segment_rows") or None) for e in range(2)]).astype(np.int32)
        d = {k: v.copy() for k, v in synthetic.init_weights(U, I, F, seed=1492).items()}
        outd = orc.fit(np.ascontiguousarray(pairs[by_csr]), np.ascontiguousarray(sw[by_csr]), csr.offsets, csr.items, x_uf, x_if,
                       d["w_i"], d["w_if"], d["v_u"], d["v_i"], d["v_uf"], d["v_if"], 0.01, 0.1, 0.1, "constant", 0.25, 1, 2,
                       perms=perms, rng_mode=orc.RNG_COUNTER, seed=1492, membership="binary", pos_step=pos_step, user_step=user_step,
                       **order.oracle_stripes(csr.offsets, 1492, range(2), geo, I))
        print("    against the DAMPED sequential oracle: LL gpu/oracle - 1 =", np.round(rep["log_likelihood"] / outd["ll"] - 1.0, 5),
              " damped / plain oracle - 1 =", np.round(outd["ll"] / out["ll"] - 1.0, 5),
              " scales: pos min %.3f mean %.3f, user min %.3f" % (pos_step.min(), pos_step.mean(), user_step.min()), flush=True)
