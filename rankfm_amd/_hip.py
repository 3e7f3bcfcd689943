"""ctypes binding of librankfm_hip.so -- the C ABI declared in include/rankfm_hip.h.

There is deliberately NO fallback: if the shared library is missing or no gfx950 device is visible, every
entry point raises.  (The CPU restatement under oracle/ is test infrastructure and is never imported here.)
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "librankfm_hip.so")

ABI_VERSION = 6
OK = 0
ERR_BAD_ARG, ERR_UNKNOWN_SCHEDULE, ERR_NO_DEVICE, ERR_HIP, ERR_UNSUPPORTED, ERR_USER_SATURATED, ERR_WORKSPACE = (
    -1, -2, -3, -4, -5, -6, -7)
ERR_NONFINITE = 100
SCHEDULE_CONSTANT, SCHEDULE_INVSCALING = 0, 1
MODE_HOGWILD, MODE_SERIAL = 0, 1
RNG_MT19937, RNG_COUNTER = 0, 1
REFERENCE_MT_SEED = 1492       # rankfm/_rankfm.pyx:182


class FitTuning(C.Structure):
    """rfm_fit_tuning: the experiments' overrides (0 = automatic everywhere); production passes none (FitConfig.tuning = NULL)"""
    _fields_ = [
        ("n_workgroups", C.c_int32), ("rows_per_launch", C.c_int32), ("debug_shape", C.c_int32), ("debug_flags", C.c_int32),
        ("segment_rows", C.c_int32), ("hot_publications", C.c_int32), ("feature_waves", C.c_int32), ("table_producers", C.c_int32),
        ("table_every", C.c_int32), ("table_step_pct", C.c_int32), ("table_batch", C.c_int32), ("hot_sweep_every", C.c_int32), ("hot_slots", C.c_int32), ("table_pace_pct", C.c_int32),
    ]


class FitConfig(C.Structure):
    _fields_ = [
        ("n_interactions", C.c_int64),
        ("n_users", C.c_int32), ("n_items", C.c_int32),
        ("n_user_features", C.c_int32), ("n_item_features", C.c_int32), ("n_factors", C.c_int32),
        ("has_user_features", C.c_int32), ("has_item_features", C.c_int32),
        ("alpha", C.c_float), ("beta", C.c_float), ("learning_rate", C.c_float),
        ("learning_schedule", C.c_int32), ("learning_exponent", C.c_float),
        ("max_samples", C.c_int32), ("epochs", C.c_int32), ("epoch_begin", C.c_int32), ("rng_epoch_offset", C.c_int32),
        ("mode", C.c_int32), ("rng", C.c_int32), ("seed", C.c_uint32),
        ("check_finite", C.c_int32), ("want_penalty", C.c_int32),
        ("hogwild_damping", C.c_float),
        ("epoch_part_index", C.c_int32), ("epoch_parts", C.c_int32), ("keep_layout", C.c_int32),
        ("plan_token", C.c_int64), ("layout_token", C.c_int64),
        ("tuning", C.POINTER(FitTuning)),
    ]


#: names of the geometry overrides of rfm_fit_tuning (0 = automatic); EngineOptions.tune / DeviceSession(tune=...) carry them
TUNE_FIELDS = ("segment_rows", "hot_publications", "feature_waves", "table_producers", "table_every", "table_step_pct", "table_batch", "table_pace_pct", "hot_sweep_every", "hot_slots")


def tune_kwargs(tune):
    """{'segment_rows': 16, ...} -> FitTuning keyword arguments (unknown names raise)"""
    out = {}
    for k, v in (tune or {}).items():
        if k not in TUNE_FIELDS:
            raise ValueError("unknown geometry override %r (known: %s)" % (k, ", ".join(TUNE_FIELDS)))
        out[k] = int(v)
    return out


def make_tuning(tune=None, n_workgroups=0, rows_per_launch=0, debug_shape=0, debug_flags=0):
    """a FitTuning for FitConfig.tuning, or None when everything is automatic (production: the library gets a NULL pointer)"""
    kw = dict(tune_kwargs(tune), n_workgroups=int(n_workgroups), rows_per_launch=int(rows_per_launch), debug_shape=int(debug_shape),
              debug_flags=int(debug_flags))
    return FitTuning(**kw) if any(kw.values()) else None


class FitBuffers(C.Structure):
    _fields_ = [
        ("interactions", C.c_void_p), ("sample_weight", C.c_void_p),
        ("csr_offsets", C.c_void_p), ("csr_items", C.c_void_p),
        ("x_uf", C.c_void_p), ("x_if", C.c_void_p),
        ("w_i", C.c_void_p), ("w_if", C.c_void_p), ("v_u", C.c_void_p), ("v_i", C.c_void_p),
        ("v_uf", C.c_void_p), ("v_if", C.c_void_p),
        ("perms", C.c_void_p), ("workspace", C.c_void_p), ("workspace_bytes", C.c_size_t),
    ]


class FitReport(C.Structure):
    _fields_ = [
        ("log_likelihood", C.POINTER(C.c_double)), ("reg_penalty", C.POINTER(C.c_double)),
        ("sgd_kernel_ms", C.POINTER(C.c_float)), ("n_draws", C.POINTER(C.c_int64)),
        ("epochs_done", C.c_int32), ("nonfinite_array", C.c_int32),
        ("launches_per_epoch", C.c_int32), ("waves_per_launch", C.c_int32),
        ("plan_token", C.c_int64),
        ("workgroups", C.c_int32), ("groups_per_workgroup", C.c_int32), ("working_groups", C.c_int64),
        ("units_per_launch", C.c_int64), ("n_units", C.c_int64),
        ("segment_rows", C.c_int32), ("table_producers", C.c_int32), ("layout_token", C.c_int64), ("table_steps", C.c_int64), ("feat_diag", C.c_int64 * 8),
        ("table_overlap_us", C.c_int64), ("table_span_us", C.c_int64 * 2), ("shader_mhz", C.c_float), ("reserved_report", C.c_int32),
    ]

    def geometry(self):
        """launch geometry as a dict (rankfm_amd.order mirrors the engine's negative draws from it)"""
        return dict(self._geometry(), feat_diag=[int(x) for x in self.feat_diag], table_overlap_us=int(self.table_overlap_us),
                    table_span_us=[int(x) for x in self.table_span_us], shader_mhz=float(self.shader_mhz))

    def _geometry(self):
        return {k: int(getattr(self, k)) for k in ("workgroups", "groups_per_workgroup", "working_groups", "units_per_launch",
                                                   "n_units", "launches_per_epoch", "segment_rows",
                                                   "table_producers", "table_steps")}


class ModelView(C.Structure):
    _fields_ = [
        ("n_users", C.c_int32), ("n_items", C.c_int32), ("n_user_features", C.c_int32),
        ("n_item_features", C.c_int32), ("n_factors", C.c_int32),
        ("has_user_features", C.c_int32), ("has_item_features", C.c_int32),
        ("x_uf", C.c_void_p), ("x_if", C.c_void_p), ("w_i", C.c_void_p), ("w_if", C.c_void_p),
        ("v_u", C.c_void_p), ("v_i", C.c_void_p), ("v_uf", C.c_void_p), ("v_if", C.c_void_p),
    ]


# every symbol include/rankfm_hip.h declares (tests check the library exports all of them)
EXPORTS = (
    "rfm_abi_version", "rfm_status_string", "rfm_last_error", "rfm_device_count", "rfm_fit_supported",
    "rfm_fit_workspace_bytes", "rfm_fit_device", "rfm_fit_host", "rfm_fit_export_weights", "rfm_predict_device", "rfm_predict_host",
    "rfm_recommend_device", "rfm_recommend_workspace_bytes", "rfm_recommend_host", "rfm_similar_host", "rfm_hbm_probe",
    "rfm_delta_begin", "rfm_delta_finish", "rfm_release_cache",
)

_lib = None


class EngineUnavailable(RuntimeError):
    """librankfm_hip.so is missing/unloadable or no MI355X is visible.  There is no CPU fallback."""


def lib():
    """load librankfm_hip.so (built by rankfm_amd._build / __graft_entry__.build)"""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise EngineUnavailable(
            "%s not found: build it with `python -m rankfm_amd._build` (hipcc, gfx950). "
            "rankfm_amd has no CPU fallback." % LIB_PATH)
    # One HIP runtime per process: PyTorch-ROCm ships its own libamdhip64.so.7 (same soname as /opt/rocm's).  Whichever
    # is loaded first serves both, and PyTorch does not work on top of a foreign one -- so let PyTorch load its runtime
    # before librankfm_hip.so binds to that soname.  (A host without PyTorch resolves it through the library's RUNPATH.)
    try:
        import torch  # noqa: F401
    except ImportError:
        pass
    try:
        L = C.CDLL(LIB_PATH)
    except OSError as e:
        raise EngineUnavailable("cannot load %s: %s" % (LIB_PATH, e))
    L.rfm_abi_version.restype = C.c_int
    L.rfm_status_string.restype = C.c_char_p
    L.rfm_status_string.argtypes = [C.c_int]
    L.rfm_last_error.restype = C.c_char_p
    L.rfm_device_count.restype = C.c_int
    L.rfm_fit_supported.restype = C.c_int
    L.rfm_fit_supported.argtypes = [C.POINTER(FitConfig)]
    L.rfm_fit_workspace_bytes.restype = C.c_size_t
    L.rfm_fit_workspace_bytes.argtypes = [C.POINTER(FitConfig)]
    L.rfm_fit_device.restype = C.c_int
    L.rfm_fit_device.argtypes = [C.POINTER(FitConfig), C.POINTER(FitBuffers), C.c_void_p, C.POINTER(FitReport)]
    L.rfm_fit_export_weights.restype = C.c_int
    L.rfm_fit_export_weights.argtypes = [C.POINTER(FitConfig), C.POINTER(FitBuffers), C.c_void_p]
    L.rfm_fit_host.restype = C.c_int
    L.rfm_fit_host.argtypes = [C.POINTER(FitConfig), C.POINTER(FitBuffers), C.c_int, C.POINTER(FitReport)]
    L.rfm_predict_device.restype = C.c_int
    L.rfm_predict_device.argtypes = [C.POINTER(ModelView), C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p]
    L.rfm_predict_host.restype = C.c_int
    L.rfm_predict_host.argtypes = [C.POINTER(ModelView), C.c_int64, C.c_void_p, C.c_void_p, C.c_int]
    L.rfm_recommend_workspace_bytes.restype = C.c_size_t
    L.rfm_recommend_workspace_bytes.argtypes = [C.POINTER(ModelView), C.c_int64, C.c_int32]
    L.rfm_recommend_device.restype = C.c_int
    L.rfm_recommend_device.argtypes = [C.POINTER(ModelView), C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32,
                                       C.c_int32, C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p]
    L.rfm_recommend_host.restype = C.c_int
    L.rfm_recommend_host.argtypes = [C.POINTER(ModelView), C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32,
                                     C.c_int32, C.c_void_p, C.c_int]
    L.rfm_delta_begin.restype = C.c_int
    L.rfm_delta_begin.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p]
    L.rfm_delta_finish.restype = C.c_int
    L.rfm_delta_finish.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_float, C.c_size_t, C.c_void_p]
    L.rfm_hbm_probe.restype = C.c_int
    L.rfm_hbm_probe.argtypes = [C.c_size_t, C.c_int, C.POINTER(C.c_double), C.POINTER(C.c_double)]
    L.rfm_release_cache.restype = None
    L.rfm_similar_host.restype = C.c_int
    L.rfm_similar_host.argtypes = [C.POINTER(ModelView), C.c_int32, C.c_int32, C.c_int32, C.c_void_p, C.c_int]
    if L.rfm_abi_version() != ABI_VERSION:
        raise EngineUnavailable("librankfm_hip.so ABI %d != binding ABI %d: rebuild" % (L.rfm_abi_version(), ABI_VERSION))
    _lib = L
    # rfm_fit_host keeps one staging allocation per device between calls (arenas above 1 GiB are freed when their call returns): give
    # what is left back before the HIP runtime goes down with the interpreter; release_cache() does it on demand
    import atexit
    atexit.register(release_cache)
    return L


def release_cache():
    """free the device staging memory the drop-in `_fit` path keeps between calls (include/rankfm_hip.h rfm_release_cache)"""
    if _lib is not None:
        _lib.rfm_release_cache()


def status_string(rc):
    return lib().rfm_status_string(int(rc)).decode()


def raise_for_status(rc):
    """map a C status onto the exception type the reference raises at the same point"""
    if rc == OK:
        return
    msg = status_string(rc)
    if rc >= ERR_NONFINITE:
        raise AssertionError(msg)                       # assert_finite, rankfm/_rankfm.pyx:95-103
    if rc == ERR_UNKNOWN_SCHEDULE:
        raise ValueError(msg)                           # rankfm/_rankfm.pyx:225
    if rc == ERR_NO_DEVICE:
        raise EngineUnavailable(msg + " (" + lib().rfm_last_error().decode() + ")")
    if rc == ERR_HIP:
        raise RuntimeError(msg + ": " + lib().rfm_last_error().decode())
    if rc in (ERR_BAD_ARG, ERR_UNSUPPORTED, ERR_USER_SATURATED, ERR_WORKSPACE):
        raise ValueError(msg)
    raise RuntimeError("rankfm_hip status %d: %s" % (rc, msg))
