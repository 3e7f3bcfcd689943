"""ctypes wrapper around oracle/librfm_oracle.so -- TEST INFRASTRUCTURE, not product code.

Only tests/, `__graft_entry__.smoke()` and the `cpu_baseline` leg of bench.py may import this
module.  It is the checker (and the timed CPU baseline), never the thing shipped or measured
as the engine.  See oracle/rfm_oracle.c for the reference file:line each function restates.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "librfm_oracle.so")
_lib = None

RNG_MT19937 = 0
RNG_COUNTER = 1
REFERENCE_MT_SEED = 1492          # rankfm/_rankfm.pyx:182

_ARRAY_NAMES = ("w_i", "w_if", "v_u", "v_i", "v_uf", "v_if")   # order of assert_finite, _rankfm.pyx:98-103


class OracleParams(C.Structure):
    _fields_ = [
        ("N", C.c_int64),
        ("U", C.c_int32), ("I", C.c_int32), ("P", C.c_int32), ("Q", C.c_int32), ("F", C.c_int32),
        ("has_uf", C.c_int32), ("has_if", C.c_int32),
        ("alpha", C.c_float), ("beta", C.c_float), ("learning_rate", C.c_float),
        ("schedule", C.c_int32),
        ("learning_exponent", C.c_float),
        ("max_samples", C.c_int32),
        ("epochs", C.c_int32), ("epoch_begin", C.c_int32),
        ("rng_mode", C.c_int32),
        ("seed", C.c_uint32),
        ("membership", C.c_int32),
        ("table_every", C.c_int32), ("table_step", C.c_float),
        ("table_head_every", C.c_int32), ("table_head_rows", C.c_int64), ("table_tail", C.c_int64), ("table_quiet_rows", C.c_int64), ("table_batch", C.c_int32),
    ]


def build(force=False):
    """compile the C restatement (gcc, seconds)"""
    src = os.path.join(_HERE, "rfm_oracle.c")
    hdr = os.path.join(_HERE, "..", "include", "rfm_rng.h")
    if (not force and os.path.exists(_LIB_PATH)
            and os.path.getmtime(_LIB_PATH) >= max(os.path.getmtime(src), os.path.getmtime(hdr))):
        return _LIB_PATH
    subprocess.check_call(["make", "-C", _HERE, "-B", "librfm_oracle.so"], stdout=subprocess.DEVNULL)
    return _LIB_PATH


def lib():
    global _lib
    if _lib is None:
        build()
        _lib = C.CDLL(_LIB_PATH)
        _lib.rfm_oracle_fit.restype = C.c_int
        _lib.rfm_oracle_fit_ex.restype = C.c_int
        _lib.rfm_oracle_reg_penalty.restype = C.c_double
    return _lib


def _p(a, ct):
    return None if a is None else a.ctypes.data_as(C.POINTER(ct))


def _f32(a):
    assert a.dtype == np.float32 and a.flags.c_contiguous, "float32 C-contiguous expected"
    return a


def user_items_to_csr(user_items, n_users):
    """dict{u -> sorted int32 array} (rankfm.py:174) -> (offsets int64[U+1], items int32[nnz])"""
    off = np.zeros(n_users + 1, dtype=np.int64)
    for u in range(n_users):
        off[u + 1] = off[u] + len(user_items[u])
    items = np.empty(int(off[-1]), dtype=np.int32)
    for u in range(n_users):
        items[off[u]:off[u + 1]] = user_items[u]
    return off, items


def mt_stream(seed, n):
    out = np.empty(n, dtype=np.uint32)
    lib().rfm_oracle_mt_stream(C.c_uint32(seed), C.c_int(n), _p(out, C.c_uint32))
    return out


def fit(interactions, sample_weight, csr_off, csr_items, x_uf, x_if, w_i, w_if, v_u, v_i, v_uf, v_if,
        alpha, beta, learning_rate, learning_schedule, learning_exponent, max_samples, epochs,
        perms=None, rng_mode=RNG_MT19937, seed=REFERENCE_MT_SEED, epoch_begin=0, membership="linear",
        has_uf=None, has_if=None, want_negatives=False, pos_step=None, user_step=None, pos_step_bias=None, neg_step=None, table_every=0, table_step=0.0,
        table_head_every=0, table_head_rows=0, table_tail=0, table_quiet_rows=0, table_batch=0):
    """Run the sequential restatement of `_fit` IN PLACE on the six weight arrays.

    `pos_step` [I] / `user_step` [U] (both or neither): NOT the reference's algorithm any more -- the engine's Hogwild step damping
    applied sequentially (rfm_oracle_fit_ex), to separate what the damping changes from what asynchrony changes.

    Returns dict(ll=float64[epochs], ll64=float64[epochs], neg=int32[epochs,N] | None, nsamp=int32[epochs,N] | None): `ll` is the
    reference's float-accumulated log-likelihood (what it prints; pinned by the golden vectors), `ll64` the same sum in double --
    at millions of rows the float accumulator rounds away every small term (~0.5 % at 5 M rows), so statistical comparisons of a
    trajectory should use `ll64`.
    Raises AssertionError like the reference's assert_finite (_rankfm.pyx:95-103) and ValueError for an
    unknown learning schedule (_rankfm.pyx:225).
    """
    if learning_schedule not in ("constant", "invscaling"):
        raise ValueError("unknown [learning_schedule]")
    N = interactions.shape[0]
    assert interactions.dtype == np.int32 and interactions.flags.c_contiguous
    assert csr_off.dtype == np.int64 and csr_items.dtype == np.int32
    U, F = v_u.shape
    I = v_i.shape[0]
    P, Q = v_uf.shape[0], v_if.shape[0]
    prm = OracleParams(
        N=N, U=U, I=I, P=P, Q=Q, F=F,
        has_uf=int(np.asarray(x_uf).any()) if has_uf is None else int(has_uf),
        has_if=int(np.asarray(x_if).any()) if has_if is None else int(has_if),
        alpha=alpha, beta=beta, learning_rate=learning_rate,
        schedule=0 if learning_schedule == "constant" else 1,
        learning_exponent=learning_exponent, max_samples=max_samples,
        epochs=epochs, epoch_begin=epoch_begin, rng_mode=rng_mode, seed=seed,
        membership=0 if membership == "linear" else 1,
        table_every=int(table_every), table_step=float(table_step),      # (analysis options, rfm_oracle.c; 0 / 0 = the reference)
        table_head_every=int(table_head_every), table_head_rows=int(table_head_rows), table_tail=int(table_tail), table_quiet_rows=int(table_quiet_rows), table_batch=int(table_batch))
    if perms is not None:
        perms = np.ascontiguousarray(perms, dtype=np.int32)
        assert perms.shape == (epochs, N)
    ll = np.zeros(epochs, dtype=np.float64)
    neg = np.full((epochs, N), -1, dtype=np.int32) if want_negatives else None
    nsamp = np.zeros((epochs, N), dtype=np.int32) if want_negatives else None
    args = [C.byref(prm), _p(interactions, C.c_int32), _p(_f32(sample_weight), C.c_float),
            _p(csr_off, C.c_int64), _p(csr_items, C.c_int32),
            _p(_f32(x_uf), C.c_float), _p(_f32(x_if), C.c_float),
            _p(_f32(w_i), C.c_float), _p(_f32(w_if), C.c_float), _p(_f32(v_u), C.c_float),
            _p(_f32(v_i), C.c_float), _p(_f32(v_uf), C.c_float), _p(_f32(v_if), C.c_float),
            _p(perms, C.c_int32), _p(ll, C.c_double), _p(neg, C.c_int32), _p(nsamp, C.c_int32)]
    ll64 = np.zeros(epochs, dtype=np.float64)
    if pos_step is not None or user_step is not None:
        pos_step = np.ascontiguousarray(pos_step, dtype=np.float32)
        user_step = np.ascontiguousarray(user_step, dtype=np.float32)
        assert pos_step.shape == (I,) and user_step.shape == (U,)
        if pos_step_bias is not None:       # (a scale of its own for the positive item's bias step)
            pos_step_bias = np.ascontiguousarray(pos_step_bias, dtype=np.float32)
            assert pos_step_bias.shape == (I,)
        if neg_step is not None:            # (the scale of the NEGATIVE item's step)
            neg_step = np.ascontiguousarray(neg_step, dtype=np.float32)
            assert neg_step.shape == (I,)
    rc = lib().rfm_oracle_fit_ex(*args, _p(pos_step, C.c_float), _p(user_step, C.c_float), _p(pos_step_bias, C.c_float), _p(neg_step, C.c_float), _p(ll64, C.c_double))
    if rc >= 100:
        raise AssertionError("[%s] are not finite" % _ARRAY_NAMES[rc - 100])
    if rc != 0:
        raise ValueError("rfm_oracle_fit: bad argument (rc=%d)" % rc)
    return dict(ll=ll, ll64=ll64, neg=neg, nsamp=nsamp)


def reg_penalty(alpha, beta, w_i, w_if, v_u, v_i, v_uf, v_if):
    U, F = v_u.shape
    return float(lib().rfm_oracle_reg_penalty(
        C.c_float(alpha), C.c_float(beta), C.c_int32(U), C.c_int32(v_i.shape[0]), C.c_int32(v_uf.shape[0]),
        C.c_int32(v_if.shape[0]), C.c_int32(F), _p(_f32(w_i), C.c_float), _p(_f32(w_if), C.c_float),
        _p(_f32(v_u), C.c_float), _p(_f32(v_i), C.c_float), _p(_f32(v_uf), C.c_float), _p(_f32(v_if), C.c_float)))


def predict(pairs_f32, x_uf, x_if, w_i, w_if, v_u, v_i, v_uf, v_if):
    pairs_f32 = np.ascontiguousarray(pairs_f32, dtype=np.float32)
    U, F = v_u.shape
    out = np.empty(pairs_f32.shape[0], dtype=np.float32)
    lib().rfm_oracle_predict(
        C.c_int64(pairs_f32.shape[0]), _p(pairs_f32, C.c_float), C.c_int32(U), C.c_int32(v_i.shape[0]),
        C.c_int32(v_uf.shape[0]), C.c_int32(v_if.shape[0]), C.c_int32(F),
        C.c_int32(int(np.asarray(x_uf).any())), C.c_int32(int(np.asarray(x_if).any())),
        _p(_f32(x_uf), C.c_float), _p(_f32(x_if), C.c_float), _p(_f32(w_i), C.c_float), _p(_f32(w_if), C.c_float),
        _p(_f32(v_u), C.c_float), _p(_f32(v_i), C.c_float), _p(_f32(v_uf), C.c_float), _p(_f32(v_if), C.c_float),
        _p(out, C.c_float))
    return out


def user_scores(u, x_uf, x_if, w_i, w_if, v_u, v_i, v_uf, v_if):
    U, F = v_u.shape
    out = np.empty(v_i.shape[0], dtype=np.float32)
    lib().rfm_oracle_user_scores(
        C.c_int32(u), C.c_int32(U), C.c_int32(v_i.shape[0]), C.c_int32(v_uf.shape[0]), C.c_int32(v_if.shape[0]),
        C.c_int32(F), C.c_int32(int(np.asarray(x_uf).any())), C.c_int32(int(np.asarray(x_if).any())),
        _p(_f32(x_uf), C.c_float), _p(_f32(x_if), C.c_float), _p(_f32(w_i), C.c_float), _p(_f32(w_if), C.c_float),
        _p(_f32(v_u), C.c_float), _p(_f32(v_i), C.c_float), _p(_f32(v_uf), C.c_float), _p(_f32(v_if), C.c_float),
        _p(out, C.c_float))
    return out


def recommend(users_f32, csr_off, csr_items, n_items, filter_previous, x_uf, x_if, w_i, w_if, v_u, v_i, v_uf, v_if):
    """_rankfm.pyx:393-460: full descending argsort per user, optional skip of seen items, first n_items.
    Returns float32 [len(users), n_items] of ITEM INDEXES (NaN rows for NaN users)."""
    users_f32 = np.asarray(users_f32, dtype=np.float32)
    rec = np.empty((len(users_f32), n_items), dtype=np.float32)
    for r, uf in enumerate(users_f32):
        if np.isnan(uf):
            rec[r] = np.nan
            continue
        u = int(uf)
        sc = user_scores(u, x_uf, x_if, w_i, w_if, v_u, v_i, v_uf, v_if)
        ranked = np.argsort(sc)[::-1]
        if filter_previous:
            seen = csr_items[csr_off[u]:csr_off[u + 1]]
            ranked = ranked[~np.isin(ranked, seen)]
        rec[r] = ranked[:n_items]
    return rec
