"""ctypes wrapper around oracle/rfm_async_sim.c -- a CPU MODEL of how the stripe kernel executes a BPR epoch (lock-step rounds of
all row groups, atomics visible at the end of a round, a workgroup's stripe and hot-row sums visible to itself at once).
TEST / ANALYSIS INFRASTRUCTURE: only tests/ and tools/ may import it.  See the header of rfm_async_sim.c for what is modelled."""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "librfm_async_sim.so")
_lib = None


class SimParams(C.Structure):
    _fields_ = [("n_rows", C.c_int64), ("U", C.c_int32), ("I", C.c_int32), ("F", C.c_int32), ("alpha", C.c_float), ("eta", C.c_float),
                ("epoch_key", C.c_uint32), ("n_groups", C.c_int32), ("gpb", C.c_int32), ("grid", C.c_int32),
                ("stripe_rows", C.c_int32), ("stripe_window", C.c_int32), ("cover", C.c_float), ("mean_view", C.c_float),
                ("n_hot", C.c_int32), ("defer", C.c_int32), ("publish_now", C.c_int32), ("phases", C.c_int32)]


def lib():
    global _lib
    if _lib is None:
        src = os.path.join(_HERE, "rfm_async_sim.c")
        hdr = os.path.join(_HERE, "..", "include", "rfm_rng.h")
        if not os.path.exists(_LIB_PATH) or os.path.getmtime(_LIB_PATH) < max(os.path.getmtime(src), os.path.getmtime(hdr)):
            subprocess.check_call(["gcc", "-O2", "-ffast-math", "-fPIC", "-Wall", "-Wno-unused-function", "-shared", "-o", _LIB_PATH, src, "-lm"])
        _lib = C.CDLL(_LIB_PATH)
        _lib.rfm_async_sim_epoch.restype = C.c_int
    return _lib


def default_geometry(n_users, n_items, n_rows, n_segments, factors=64, stripes=True, window_factor=8.0, single_group=False, segment_rows=None):
    """the launch geometry rfm_api.hip plans for a BPR problem without features on a 256-CU device (16 wavefronts of four 16-lane
    groups per workgroup): what DeviceSession.geometry() reports, restated so that the model runs without a GPU"""
    gpb, cus = 64, 256
    cap = min(cus, (n_rows // 128 + gpb - 1) // gpb, (min(n_users, n_items) // 3 + gpb - 1) // gpb)
    need = (n_segments + gpb - 1) // gpb
    grid = max(1, min(need, cap))
    max_groups = max(1, min(n_rows // 128, min(n_users, n_items) // 3))
    working = min(grid * gpb, max_groups)
    g = dict(workgroups=grid, groups_per_workgroup=gpb, working_groups=working, units_per_launch=n_segments, n_units=n_segments,
             stripe_rows=0, stripe_window=1, single_group=single_group, epoch_part=None, n_items=n_items, segment_rows=segment_rows)
    if stripes and min(n_rows // 128, min(n_users, n_items) // 3) >= 32 * 16 * 4:
        window = int(max(1, min(32, int(window_factor * n_items / working + 0.5))))
        rows = min(256, (156 * 1024 - 4 * (factors + 1)) // (4 * (1 + 2 * (factors + 1))), n_items)
        rows = max(1, min(rows, max(16, min(working, gpb) * window // 2), n_items // grid))
        if n_items >= 2 * rows:
            while rows > 1:
                step = (grid * rows) % n_items
                if rows <= step <= n_items - rows:
                    break
                rows -= 1
        g.update(stripe_rows=rows, stripe_window=window)
    return g


def damping_plan(item_counts, csr_offsets, geometry, damping=128.0, hot=True, factors=64):
    """the engine's step damping and hot-row choice (rfm_api.hip "plan", parts 2 and 3) for `geometry`:
    (pos_step [I], user_step [U], hot_slot [I] (-1 = none), hot_period [n_hot])"""
    cnt = np.asarray(item_counts, dtype=np.float64)
    n_rows, n_items = cnt.sum(), len(cnt)
    in_flight = float(geometry["working_groups"])
    grid = float(geometry["workgroups"])
    pos = np.minimum(1.0, damping / np.maximum(in_flight * cnt / n_rows, 1e-30))
    hot_slot = np.full(n_items, -1, dtype=np.int32)
    periods = []
    if hot:
        g0 = 256 * 16.0 * 4
        order_ = [int(i) for i in np.argsort(-cnt, kind="stable") if cnt[i] * g0 / n_rows >= 16.0][:min(64, 12288 // (factors + 2))]
        for s, i in enumerate(order_):
            period = int(min(64, max(1, int(cnt[i] / (grid * 48.0) + 0.5))))
            n = in_flight * cnt[i] / n_rows + 0.5 * grid * period
            pos[i] = min(1.0, damping / n)
            hot_slot[i] = s
            periods.append(period)
    deg = np.diff(np.asarray(csr_offsets))
    user = np.minimum(1.0, (damping * float(geometry["n_units"]) / in_flight) / np.maximum(deg, 1))
    return pos.astype(np.float32), user.astype(np.float32), hot_slot, np.asarray(periods or [1], dtype=np.int32)


def epoch(pairs_csr, sample_weight_csr, csr_offsets, csr_items, weights, seed, epoch_index, geometry, alpha=0.01, eta=0.1,
          pos_step=None, user_step=None, hot_slot=None, hot_period=None, mean_view=1.0, defer=True, skew=None, phases=1, publish_now=False):
    """one epoch of the model IN PLACE on weights["w_i"], ["v_u"], ["v_i"]; returns (log-likelihood, rows that left their stripe)"""
    from rankfm_amd import order
    geometry = dict(geometry)
    sch = order.row_schedule(csr_offsets, seed, epoch_index, geometry)
    n = len(sch["pos"])
    # `skew` [workgroups]: rounds by which a workgroup lags the others (the real kernel's workgroups are not in lock-step; their
    # window schedule depends on their OWN iteration count, so lagging workgroups hold their stripes while others have moved on)
    rounds = sch["it"] if skew is None else sch["it"] + np.asarray(skew, dtype=np.int64)[sch["workgroup"]]
    by_round = np.lexsort((sch["group"], rounds))
    flags = ((sch["t"] == 0).astype(np.int32) | ((sch["t"] == sch["seg_len"][sch["sp"]] - 1).astype(np.int32) << 1))
    R = int(geometry["stripe_rows"])
    stripe = order.row_stripes(csr_offsets, seed, epoch_index, geometry)[sch["pos"]] if R > 0 else np.zeros(n, np.int32)
    if R > 0 and phases > 1:
        # experiment: the stripes of window w tile part (w % phases) of the item permutation, so that workgroups up to phases - 1
        # windows apart never hold the same item
        I_ = int(len(weights["w_i"]))
        window = sch["it"] // int(geometry["stripe_window"])
        part = window % phases
        lo = (part * I_ + phases - 1) // phases
        hi = ((part + 1) * I_ + phases - 1) // phases
        salt = int(order.mix32(np.uint64(int(sch["epoch_key"]) ^ 0x68E31DA4)))
        slot = (window // phases) * int(geometry["workgroups"]) + sch["workgroup"]
        stripe = (lo + (slot * R + salt) % (hi - lo)).astype(np.int32)
    a32 = lambda x: np.ascontiguousarray(x[by_round], dtype=np.int32)
    U, F = weights["v_u"].shape
    I = len(weights["w_i"])
    n_hot = int(hot_slot.max()) + 1 if hot_slot is not None and hot_slot.max() >= 0 else 0
    grid = int(geometry["workgroups"])
    prm = SimParams(n_rows=n, U=U, I=I, F=F, alpha=alpha, eta=eta, epoch_key=int(sch["epoch_key"]),
                    n_groups=int(geometry["working_groups"]), gpb=int(geometry["groups_per_workgroup"]), grid=grid,
                    stripe_rows=R, stripe_window=int(geometry["stripe_window"]), cover=min(1.0, grid * R / I) if R else 0.0,
                    mean_view=float(mean_view), n_hot=n_hot, defer=int(bool(defer)), publish_now=int(bool(publish_now)), phases=int(phases))
    P = lambda a, ct: None if a is None else a.ctypes.data_as(C.POINTER(ct))
    rp, rg, rr, rf, rs = a32(sch["pos"]), a32(sch["group"]), a32(rounds), a32(flags), a32(stripe)
    ll, fb = C.c_double(0.0), C.c_int64(0)
    keep = [np.ascontiguousarray(x, dtype=np.float32) if x is not None else None for x in (pos_step, user_step)]
    hs = np.ascontiguousarray(hot_slot, dtype=np.int32) if n_hot else None
    hp = np.ascontiguousarray(hot_period, dtype=np.int32) if n_hot else None
    rc = lib().rfm_async_sim_epoch(C.byref(prm), P(pairs_csr, C.c_int32), P(sample_weight_csr, C.c_float), P(csr_offsets, C.c_int64),
                                   P(csr_items, C.c_int32), P(rp, C.c_int32), P(rg, C.c_int32), P(rr, C.c_int32), P(rf, C.c_int32),
                                   P(rs, C.c_int32), P(keep[0], C.c_float), P(keep[1], C.c_float), P(hs, C.c_int32), P(hp, C.c_int32),
                                   P(weights["w_i"], C.c_float), P(weights["v_u"], C.c_float), P(weights["v_i"], C.c_float),
                                   C.byref(ll), C.byref(fb))
    assert rc == 0
    return ll.value, fb.value
