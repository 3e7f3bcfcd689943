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
