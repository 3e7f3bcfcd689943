"""The reference's private operator boundary, re-hosted on the MI355X engine.

    from rankfm_amd._rankfm import _fit, _predict, _recommend

mirrors `from rankfm._rankfm import _fit, _predict, _recommend` (rankfm/rankfm.py:8): same positional
arguments, same in-place mutation of the six weight arrays, same exception types.  Each function forwards
to the C ABI in include/rankfm_hip.h (host-pointer entry points), which is what a maintainer of the
reference would bind instead of the Cython module.

Engine behaviour that the reference cannot express is carried by `EngineOptions` (module default
`DEFAULT_ENGINE`, or the keyword-only `engine=` argument).
"""
import ctypes as C
import os
from collections.abc import Mapping
from dataclasses import dataclass, field
from typing import Optional

import numpy as np

from . import _hip


@dataclass
class EngineOptions:
    """how `_fit` runs on the device

    mode      'hogwild'  thousands of wavefronts, fp32 atomic updates (production)
              'serial'   one wavefront, the reference's sequential semantics (parity / debugging)
    rng       'counter'  counter-based negative draws (include/rfm_rng.h)
              'mt19937'  the reference's MT19937 stream, seed 1492 (serial mode only)
    shuffle   'device'   keyed bijection evaluated on the fly (no index array, no host RNG)
              'numpy'    the reference's cumulative np.random.shuffle of arange(N) (rankfm/_rankfm.pyx:197,227),
                         consuming numpy's global RNG exactly like the reference
    seed      counter seed; None = one draw from numpy's global RNG per call (so np.random.seed() governs
              reproducibility as it does for the reference)
    """
    mode: str = "hogwild"
    rng: str = "counter"
    shuffle: str = "device"
    seed: Optional[int] = None
    device: Optional[int] = None
    check_finite: bool = True
    damping: float = 0.0          # hogwild step damping M (include/rankfm_hip.h: hogwild_damping); 0 default, < 0 off
    # experiments / tests only (rfm_fit_tuning; production passes a NULL tuning pointer):
    n_workgroups: int = 0
    rows_per_launch: int = 0
    debug_flags: int = 0          # include/rankfm_hip.h: bit 0 = Hogwild kernel on one row group, bit 1 = L1-bypassing loads
    tune: dict = field(default_factory=dict)   # geometry overrides (_hip.TUNE_FIELDS)

    def validated(self):
        if self.mode not in ("hogwild", "serial"):
            raise ValueError("engine mode must be 'hogwild' or 'serial'")
        if self.rng not in ("counter", "mt19937"):
            raise ValueError("engine rng must be 'counter' or 'mt19937'")
        if self.shuffle not in ("device", "numpy"):
            raise ValueError("engine shuffle must be 'device' or 'numpy'")
        if self.rng == "mt19937" and (self.mode != "serial" or self.shuffle != "numpy"):
            raise ValueError("rng='mt19937' is one serial stream: it needs mode='serial' and shuffle='numpy'")
        return self


#: the reference's exact behaviour (sequential order, MT19937 seed 1492, numpy shuffle) on one wavefront
REFERENCE_ENGINE = EngineOptions(mode="serial", rng="mt19937", shuffle="numpy")
#: production default
DEFAULT_ENGINE = EngineOptions()


def default_device():
    for key in ("RANKFM_DEVICE", "LOCAL_RANK"):
        if os.environ.get(key, "") != "":
            return int(os.environ[key])
    return 0


def _buffer(a, dtype, ndim, name):
    """the checks Cython's typed memoryviews perform at the reference boundary (ValueError on mismatch)"""
    if not isinstance(a, np.ndarray):
        raise TypeError("[%s] must be a numpy array" % name)
    if a.dtype != dtype:
        raise ValueError("Buffer dtype mismatch for [%s], expected '%s' but got '%s'" % (name, np.dtype(dtype), a.dtype))
    if a.ndim != ndim:
        raise ValueError("Buffer has wrong number of dimensions for [%s] (expected %d, got %d)" % (name, ndim, a.ndim))
    if not a.flags.c_contiguous:
        raise ValueError("ndarray [%s] is not C-contiguous" % name)
    return a


_KEY_LIMIT = 2 ** 62          # largest user * n_items + item key UserItemsCSR.from_pairs sorts as one int64


class UserItemsCSR(Mapping):
    """dict-like view {user index -> sorted int32 item indexes} over a CSR pair.

    Stands in for the reference's `user_items` dict (rankfm/rankfm.py:174) without materialising one Python
    object per user; `offsets` is int64 [U+1], `items` int32 [nnz] sorted within each user.
    """

    def __init__(self, offsets, items):
        self.offsets = np.ascontiguousarray(offsets, dtype=np.int64)
        self.items = np.ascontiguousarray(items, dtype=np.int32)

    def __getitem__(self, u):
        u = int(u)
        if u < 0 or u >= len(self.offsets) - 1:
            raise KeyError(u)
        return self.items[self.offsets[u]:self.offsets[u + 1]]

    def __iter__(self):
        return iter(range(len(self.offsets) - 1))

    def __len__(self):
        return len(self.offsets) - 1

    @classmethod
    def from_pairs(cls, user_idx, item_idx, n_users):
        """sorted-unique... NOT unique: the reference keeps duplicates (rankfm/rankfm.py:174 sorts, does not dedupe)"""
        user_idx = np.asarray(user_idx, dtype=np.int64)
        item_idx = np.asarray(item_idx, dtype=np.int64)
        counts = np.bincount(user_idx, minlength=n_users)
        off = np.zeros(n_users + 1, dtype=np.int64)
        np.cumsum(counts, out=off[1:])
        bound = int(item_idx.max()) + 1 if len(item_idx) else 1
        if n_users * bound < _KEY_LIMIT:
            # one int64 key per pair and a plain sort: ~10x faster than lexsort at 5 M pairs (numpy sorts int64 with SIMD)
            key = user_idx * bound + item_idx
            key.sort()
            return cls(off, (key % bound).astype(np.int32))
        order = np.lexsort((item_idx, user_idx))
        return cls(off, item_idx[order].astype(np.int32))

    @classmethod
    def from_mapping(cls, user_items, n_users):
        if isinstance(user_items, UserItemsCSR):
            return user_items
        lens = np.fromiter((len(user_items[u]) for u in range(n_users)), dtype=np.int64, count=n_users)
        off = np.zeros(n_users + 1, dtype=np.int64)
        np.cumsum(lens, out=off[1:])
        items = (np.concatenate([np.asarray(user_items[u], dtype=np.int32) for u in range(n_users)])
                 if n_users and off[-1] else np.zeros(0, dtype=np.int32))
        return cls(off, items)


def _ptr(a):
    return None if a is None else a.ctypes.data


def _model_view(x_uf, x_if, w_i, w_if, v_u, v_i, v_uf, v_if):
    _buffer(x_uf, np.float32, 2, "x_uf"); _buffer(x_if, np.float32, 2, "x_if")
    _buffer(w_i, np.float32, 1, "w_i"); _buffer(w_if, np.float32, 1, "w_if")
    _buffer(v_u, np.float32, 2, "v_u"); _buffer(v_i, np.float32, 2, "v_i")
    _buffer(v_uf, np.float32, 2, "v_uf"); _buffer(v_if, np.float32, 2, "v_if")
    return _hip.ModelView(
        n_users=v_u.shape[0], n_items=v_i.shape[0], n_user_features=v_uf.shape[0], n_item_features=v_if.shape[0],
        n_factors=v_u.shape[1],
        has_user_features=int(x_uf.any()), has_item_features=int(x_if.any()),      # rankfm/_rankfm.pyx:193-194
        x_uf=_ptr(x_uf), x_if=_ptr(x_if), w_i=_ptr(w_i), w_if=_ptr(w_if), v_u=_ptr(v_u), v_i=_ptr(v_i),
        v_uf=_ptr(v_uf), v_if=_ptr(v_if))


def numpy_epoch_permutations(n_rows, epochs):
    """the reference's shuffle: ONE int32 arange(N) shuffled in place once per epoch with numpy's global RNG
    (rankfm/_rankfm.pyx:197,227) -- cumulative, so epoch e's order is a shuffle of epoch e-1's"""
    idx = np.arange(n_rows, dtype=np.int32)
    perms = np.empty((epochs, n_rows), dtype=np.int32)
    for e in range(epochs):
        np.random.shuffle(idx)
        perms[e] = idx
    return perms


def _fit(interactions, sample_weight, user_items, x_uf, x_if, w_i, w_if, v_u, v_i, v_uf, v_if,
         alpha, beta, learning_rate, learning_schedule, learning_exponent, max_samples, epochs, verbose,
         *, engine=None, epoch_begin=0, rng_epoch_offset=0, report=None):
    """drop-in for rankfm._rankfm._fit (rankfm/_rankfm.pyx:122-142): trains IN PLACE, returns None.

    `report`, when a dict, receives per-epoch 'log_likelihood', 'reg_penalty', 'sgd_kernel_ms', 'n_draws'.
    `rng_epoch_offset` (epochs trained by earlier calls) shifts the epoch index that keys the counter RNG and the keyed
    visiting order, so that a resumed fit with a fixed engine seed does not replay the first call's order and draws; the
    learning-rate schedule still restarts at `epoch_begin` (0) like the reference's (rankfm/_rankfm.pyx:218-223).
    """
    opt = (engine or DEFAULT_ENGINE).validated()
    _buffer(interactions, np.int32, 2, "interactions")
    _buffer(sample_weight, np.float32, 1, "sample_weight")
    mv = _model_view(x_uf, x_if, w_i, w_if, v_u, v_i, v_uf, v_if)
    if learning_schedule not in ("constant", "invscaling"):
        raise ValueError("unknown [learning_schedule]")                             # rankfm/_rankfm.pyx:225
    N = interactions.shape[0]
    epochs = int(epochs)
    csr = UserItemsCSR.from_mapping(user_items, mv.n_users)

    perms = numpy_epoch_permutations(N, epochs) if opt.shuffle == "numpy" else None
    if opt.rng == "mt19937":
        seed = _hip.REFERENCE_MT_SEED if opt.seed is None else int(opt.seed)
    else:
        seed = int(np.random.randint(0, 2**31 - 1)) if opt.seed is None else int(opt.seed)

    cfg = _hip.FitConfig(
        n_interactions=N, n_users=mv.n_users, n_items=mv.n_items, n_user_features=mv.n_user_features,
        n_item_features=mv.n_item_features, n_factors=mv.n_factors,
        has_user_features=mv.has_user_features, has_item_features=mv.has_item_features,
        alpha=alpha, beta=beta, learning_rate=learning_rate,
        learning_schedule=_hip.SCHEDULE_CONSTANT if learning_schedule == "constant" else _hip.SCHEDULE_INVSCALING,
        learning_exponent=learning_exponent, max_samples=int(max_samples), epochs=epochs, epoch_begin=int(epoch_begin),
        rng_epoch_offset=int(rng_epoch_offset), mode=_hip.MODE_SERIAL if opt.mode == "serial" else _hip.MODE_HOGWILD,
        rng=_hip.RNG_MT19937 if opt.rng == "mt19937" else _hip.RNG_COUNTER, seed=seed & 0xFFFFFFFF,
        check_finite=int(opt.check_finite), want_penalty=int(bool(verbose) or report is not None),
        hogwild_damping=float(opt.damping))
    tuning = _hip.make_tuning(opt.tune, n_workgroups=opt.n_workgroups, rows_per_launch=opt.rows_per_launch, debug_flags=opt.debug_flags)
    if tuning is not None:
        cfg.tuning = C.pointer(tuning)
    buf = _hip.FitBuffers(
        interactions=_ptr(interactions), sample_weight=_ptr(sample_weight),
        csr_offsets=_ptr(csr.offsets), csr_items=_ptr(csr.items), x_uf=_ptr(x_uf), x_if=_ptr(x_if),
        w_i=_ptr(w_i), w_if=_ptr(w_if), v_u=_ptr(v_u), v_i=_ptr(v_i), v_uf=_ptr(v_uf), v_if=_ptr(v_if),
        perms=_ptr(perms), workspace=None, workspace_bytes=0)
    ll = np.zeros(epochs, dtype=np.float64)
    pen = np.zeros(epochs, dtype=np.float64)
    ms = np.zeros(epochs, dtype=np.float32)
    draws = np.zeros(epochs, dtype=np.int64)
    rep = _hip.FitReport(
        log_likelihood=ll.ctypes.data_as(C.POINTER(C.c_double)), reg_penalty=pen.ctypes.data_as(C.POINTER(C.c_double)),
        sgd_kernel_ms=ms.ctypes.data_as(C.POINTER(C.c_float)), n_draws=draws.ctypes.data_as(C.POINTER(C.c_int64)))
    device = default_device() if opt.device is None else int(opt.device)
    rc = _hip.lib().rfm_fit_host(C.byref(cfg), C.byref(buf), device, C.byref(rep))
    if verbose:
        # rankfm/_rankfm.pyx:332-336: printed after each completed epoch (the LL variable there is a C float)
        for e in range(rep.epochs_done if rc != _hip.OK else epochs):
            print("\ntraining epoch:", epoch_begin + e)
            print("log likelihood:", float(np.float32(round(float(ll[e]) - float(pen[e]), 2))))
    if report is not None:
        report.update(log_likelihood=ll, reg_penalty=pen, sgd_kernel_ms=ms, n_draws=draws, seed=seed,
                      epochs_done=rep.epochs_done, launches_per_epoch=rep.launches_per_epoch,
                      waves_per_launch=rep.waves_per_launch,
                      geometry=dict(rep.geometry(), single_group=bool(int(opt.debug_flags) & 1), seed=seed & 0xFFFFFFFF, epoch_part=None))
    _hip.raise_for_status(rc)
    return None


def _predict(pairs, x_uf, x_if, w_i, w_if, v_u, v_i, v_uf, v_if, *, device=None):
    """drop-in for rankfm._rankfm._predict (rankfm/_rankfm.pyx:345-390): float32 index pairs (NaN = unknown id)
    -> float32 scores (NaN where either index is NaN)"""
    _buffer(pairs, np.float32, 2, "pairs")
    mv = _model_view(x_uf, x_if, w_i, w_if, v_u, v_i, v_uf, v_if)
    scores = np.empty(pairs.shape[0], dtype=np.float32)
    rc = _hip.lib().rfm_predict_host(C.byref(mv), pairs.shape[0], _ptr(pairs), _ptr(scores),
                                     default_device() if device is None else int(device))
    _hip.raise_for_status(rc)
    return scores


def _recommend(users, user_items, n_items, filter_previous, x_uf, x_if, w_i, w_if, v_u, v_i, v_uf, v_if, *, device=None):
    """drop-in for rankfm._rankfm._recommend (rankfm/_rankfm.pyx:393-460): float32 user indexes (NaN = unknown)
    -> float32 [len(users), n_items] item indexes by descending utility (NaN rows for unknown users)"""
    _buffer(users, np.float32, 1, "users")
    mv = _model_view(x_uf, x_if, w_i, w_if, v_u, v_i, v_uf, v_if)
    csr = UserItemsCSR.from_mapping(user_items, mv.n_users)
    n_items = int(n_items)
    if n_items < 1 or n_items > mv.n_items:
        raise ValueError("[n_items] must be between 1 and the number of training items")
    rec = np.empty((users.shape[0], n_items), dtype=np.float32)
    rc = _hip.lib().rfm_recommend_host(C.byref(mv), users.shape[0], _ptr(users), _ptr(csr.offsets), _ptr(csr.items),
                                       n_items, int(bool(filter_previous)), _ptr(rec),
                                       default_device() if device is None else int(device))
    _hip.raise_for_status(rc)
    return rec


def _similar(kind, index, n, x_uf, x_if, w_i, w_if, v_u, v_i, v_uf, v_if, *, device=None):
    """device side of `similar_items` (kind 0) / `similar_users` (kind 1), rankfm/rankfm.py:405-454: float32 [n] row indexes
    by descending dot product of the latent representations, the query row excluded"""
    mv = _model_view(x_uf, x_if, w_i, w_if, v_u, v_i, v_uf, v_if)
    out = np.empty(int(n), dtype=np.float32)
    rc = _hip.lib().rfm_similar_host(C.byref(mv), int(kind), int(index), int(n), _ptr(out),
                                     default_device() if device is None else int(device))
    _hip.raise_for_status(rc)
    return out
