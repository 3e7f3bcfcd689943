"""Host-side mirror (numpy) of the engine's keyed visiting order -- the spec in include/rfm_rng.h and the segment rule of
rankfm_amd/csrc/rfm_api.hip, restated so that a host program can reproduce, row for row, the order in which the
Hogwild segments kernel walks an epoch.  The parity tests use it to hand the sequential CPU oracle the very same order.
"""
import numpy as np

SEGMENT_ROWS = 32          # kSegmentRows in rfm_sgd.hpp
_M32 = np.uint64(0xFFFFFFFF)


def mix32(x):
    x = np.asarray(x, dtype=np.uint64) & _M32
    x ^= x >> np.uint64(16)
    x = (x * np.uint64(0x7feb352d)) & _M32
    x ^= x >> np.uint64(15)
    x = (x * np.uint64(0x846ca68b)) & _M32
    x ^= x >> np.uint64(16)
    return x


def epoch_key(seed, epoch):
    return int(mix32(np.uint64((int(seed) ^ ((0x9E3779B9 * (int(epoch) + 1)) & 0xFFFFFFFF)) & 0xFFFFFFFF)))


def perm_bits(n):
    b = 2
    while b < 32 and (1 << b) < n:
        b += 1
    return b


def perm(pos, n, bits, key):
    """rfm_perm for arrays: `pos`, `n`, `bits`, `key` broadcast against each other (per-element domains allowed)"""
    pos, n, bits, key = np.broadcast_arrays(np.asarray(pos, np.uint64), np.asarray(n, np.uint64), np.asarray(bits, np.uint64),
                                            np.asarray(key, np.uint64))
    mask = (np.uint64(1) << bits) - np.uint64(1)
    s1 = (bits + np.uint64(1)) >> np.uint64(1)
    s2 = np.maximum(bits >> np.uint64(1), np.uint64(1))
    ks = [mix32(key ^ np.uint64(c)) for c in (0xA511E9B3, 0x1B873593, 0xCC9E2D51, 0x38B34AE5)]
    muls = (0x9E3779B1, 0x85EBCA77, 0xC2B2AE3D, 0x27D4EB2F)
    x = pos.copy()
    todo = np.ones(x.shape, dtype=bool)
    while todo.any():
        y = x[todo]
        m, a1, a2 = mask[todo], s1[todo], s2[todo]
        for r in range(4):
            y = (y * np.uint64(muls[r]) + ks[r][todo]) & m
            y ^= y >> (a1 if r % 2 == 0 else a2)
        x[todo] = y
        todo = x >= n
    return x.astype(np.int64)


def segments(csr_offsets, rows=None):
    """(user, first CSR position, length) of every segment, in the planner's enumeration order; `rows` = longest segment of the
    plan (rfm_fit_report.segment_rows; SEGMENT_ROWS when not given)"""
    rows = int(rows or SEGMENT_ROWS)
    off = np.asarray(csr_offsets, dtype=np.int64)
    deg = np.diff(off)
    parts = (deg + rows - 1) // rows
    users = np.repeat(np.arange(len(deg), dtype=np.int64), parts)
    p = np.arange(parts.sum(), dtype=np.int64) - np.repeat(np.cumsum(parts) - parts, parts)
    d, n = deg[users], parts[users]
    s0 = off[users] + d * p // n
    s1 = off[users] + d * (p + 1) // n
    return users, s0, s1 - s0


def epoch_positions(csr_offsets, seed, epoch, segment_rows=None):
    """CSR positions in the order the segments kernel visits them in `epoch` (single-group / sequential order)"""
    return _visit(csr_offsets, seed, epoch, segment_rows)[0]


def _visit(csr_offsets, seed, epoch, segment_rows=None):
    """per visited row, in the sequential (segment order, then within-segment) enumeration of `epoch_positions`:
    (CSR position, segment-order position sp of its segment, index t inside the segment), and the segment lengths by sp"""
    users, begin, length = segments(csr_offsets, segment_rows)
    S = len(users)
    ek = epoch_key(seed, epoch)
    seg_order = perm(np.arange(S), S, perm_bits(S), ek ^ 0x5bd1e995)
    b, l = begin[seg_order], length[seg_order]
    seg_key = mix32(np.uint64(ek) ^ ((seg_order.astype(np.uint64) * np.uint64(0x9E3779B9) + np.uint64(0x7F4A7C15)) & _M32))
    t = np.arange(l.sum(), dtype=np.int64) - np.repeat(np.cumsum(l) - l, l)
    lr = np.repeat(l, l)
    bits = np.array([perm_bits(int(v)) for v in range(SEGMENT_ROWS + 1)], dtype=np.uint64)[lr]
    within = perm(t, lr, bits, np.repeat(seg_key, l))
    return np.repeat(b, l) + within, np.repeat(np.arange(S, dtype=np.int64), l), t, l, ek


def row_schedule(csr_offsets, seed, epoch, geometry):
    """Where and when the launch described by `geometry` processes every row of `epoch`: per visited row (in the enumeration of
    `epoch_positions`) its CSR position `pos`, segment-order position `sp`, index `t` in its segment, and -- restating the walk of
    sgd_segments_kernel -- the row group that processes it (`group`: group g of a launch walks the segments at order positions
    p0 + g, p0 + g + n_groups, ...), the iteration of that group's loop at which it does (`it`: rows the group has done in the
    launch before), its workgroup and launch; plus `seg_len` by sp and the epoch key."""
    pos, sp, t, seg_len, ek = _visit(csr_offsets, seed, epoch, geometry.get("segment_rows"))
    S = len(seg_len)
    single = bool(geometry["single_group"])
    n_groups = 1 if single else int(geometry["working_groups"])
    gpw = int(geometry["groups_per_workgroup"])
    upl = int(geometry["units_per_launch"])
    u_begin, u_end = 0, S
    if geometry.get("epoch_part"):
        k, n = geometry["epoch_part"]
        u_begin, u_end = S * k // n, S * (k + 1) // n
    start_iter = np.zeros(S, dtype=np.int64)       # rows its group has done in the launch before the segment
    launch_of = np.full(S, -1, dtype=np.int64)
    group_of = np.zeros(S, dtype=np.int64)
    for launch, p0 in enumerate(range(u_begin, u_end, upl)):
        p1 = min(p0 + upl, u_end)
        k = np.arange(p1 - p0, dtype=np.int64)
        stride = 1 if single else n_groups
        rounds = (len(k) + stride - 1) // stride
        m = np.zeros(rounds * stride, dtype=np.int64)
        m[:len(k)] = seg_len[p0:p1]
        m = m.reshape(rounds, stride)
        before = (np.cumsum(m, axis=0) - m).reshape(-1)[:len(k)]
        if single:                                 # one group walks the launch's segments one after the other
            before = np.cumsum(seg_len[p0:p1]) - seg_len[p0:p1]
        start_iter[p0:p1] = before
        launch_of[p0:p1] = launch
        group_of[p0:p1] = 0 if single else k % n_groups
    return dict(pos=pos, sp=sp, t=t, seg_len=seg_len, epoch_key=ek, it=start_iter[sp] + t, group=group_of[sp],
                workgroup=group_of[sp] // gpw, launch=launch_of[sp])


def row_stripes(csr_offsets, seed, epoch, geometry):
    """int32 [N]: the first position of the negative stripe (include/rfm_rng.h) every CSR position draws from in `epoch`, for the launch geometry the
    engine reported (DeviceSession.geometry() / the `geometry` entry of `_fit`'s report).  Restates the schedule of
    sgd_segments_kernel<STRIPE>: the k-th row a group processes in a launch (row_schedule) lies in window k // stripe_window of
    its workgroup; the stripe of (workgroup, window) starts at rfm_stripe_start."""
    R, RW = geometry["stripe_rows"], geometry["stripe_window"]
    n_items = int(geometry["n_items"])
    sch = row_schedule(csr_offsets, seed, epoch, geometry)
    grid = 1 if geometry["single_group"] else int(geometry["workgroups"])
    window = sch["it"] // RW
    salt = mix32((np.uint64(sch["epoch_key"]) ^ ((np.uint64(0x68E31DA4) + sch["launch"].astype(np.uint64)) & _M32)) & _M32)
    stripe = ((window * grid + sch["workgroup"]).astype(np.uint64) * np.uint64(R) + salt) % np.uint64(n_items)       # rfm_stripe_start
    out = np.zeros(len(sch["pos"]), dtype=np.int32)
    out[sch["pos"]] = stripe.astype(np.int32)
    return out


def oracle_stripes(csr_offsets, seed, epochs, geometry, n_items):
    """extra keyword arguments for oracle.fit that make the sequential oracle draw its negatives exactly like the launch
    described by `geometry` (DeviceSession.geometry()); {} when the engine draws over the whole catalogue.  `epochs` is an
    iterable of (absolute) epoch indexes."""
    if not geometry or not geometry.get("stripe_rows"):
        return {}
    g = dict(geometry, n_items=n_items)
    return dict(row_stripe=np.stack([row_stripes(csr_offsets, seed, e, g) for e in epochs]), stripe_rows=int(g["stripe_rows"]))
