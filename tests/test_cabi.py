"""The C-ABI library: loads on a CPU-only box, exports every symbol include/rankfm_hip.h declares, its structs have the
layout the ctypes binding assumes, validation works without a device, and compute entry points refuse loudly without a GPU."""
import ctypes as C
import os
import re
import subprocess

import numpy as np
import pytest

from conftest import ROOT

from rankfm_amd import _build, _hip

HEADER = os.path.join(ROOT, "include", "rankfm_hip.h")


@pytest.fixture(scope="module")
def lib():
    _build.build()
    return _hip.lib()


def test_exports_every_declared_symbol(lib):
    text = re.sub(r"/\*.*?\*/", "", open(HEADER).read(), flags=re.S)
    declared = set(re.findall(r"\b(rfm_[a-z_0-9]+)\s*\(", text))
    assert declared == set(_hip.EXPORTS), declared ^ set(_hip.EXPORTS)
    nm = subprocess.run(["nm", "-D", "--defined-only", _hip.LIB_PATH], capture_output=True, text=True, check=True).stdout
    exported = set(re.findall(r"\bT (rfm_[a-z_0-9]+)", nm))
    assert declared <= exported, declared - exported
    for name in declared:
        assert hasattr(lib, name)
    assert lib.rfm_abi_version() == _hip.ABI_VERSION


def test_struct_layouts_match_the_header(tmp_path):
    """compile the header with gcc and compare sizeof/offsetof with the ctypes mirror"""
    structs = {"rfm_fit_config": _hip.FitConfig, "rfm_fit_tuning": _hip.FitTuning, "rfm_fit_buffers": _hip.FitBuffers,
               "rfm_fit_report": _hip.FitReport, "rfm_model_view": _hip.ModelView}
    lines = ['#include <stdio.h>', '#include <stddef.h>', '#include "rankfm_hip.h"', 'int main(void){']
    for cname, ct in structs.items():
        lines.append('printf("%s %%zu\\n", sizeof(%s));' % (cname, cname))
        for f, _ in ct._fields_:
            lines.append('printf("%s.%s %%zu\\n", offsetof(%s, %s));' % (cname, f, cname, f))
    lines.append("return 0;}")
    src = tmp_path / "layout.c"
    src.write_text("\n".join(lines))
    exe = tmp_path / "layout"
    subprocess.check_call(["gcc", "-I", os.path.join(ROOT, "include"), str(src), "-o", str(exe)])
    got = dict(l.split() for l in subprocess.check_output([str(exe)], text=True).splitlines())
    for cname, ct in structs.items():
        assert int(got[cname]) == C.sizeof(ct), cname
        for f, _ in ct._fields_:
            assert int(got["%s.%s" % (cname, f)]) == getattr(ct, f).offset, (cname, f)


def _cfg(**kw):
    base = dict(n_interactions=100, n_users=10, n_items=20, n_user_features=1, n_item_features=1, n_factors=8,
                alpha=0.01, beta=0.1, learning_rate=0.1, learning_schedule=0, learning_exponent=0.25, max_samples=1,
                epochs=1, mode=_hip.MODE_HOGWILD, rng=_hip.RNG_COUNTER, seed=1)
    base.update(kw)
    return _hip.FitConfig(**base)


def test_validation_needs_no_device(lib):
    assert lib.rfm_fit_supported(C.byref(_cfg())) == _hip.OK
    assert lib.rfm_fit_workspace_bytes(C.byref(_cfg())) > 0
    assert lib.rfm_fit_supported(C.byref(_cfg(learning_schedule=7))) == _hip.ERR_UNKNOWN_SCHEDULE
    assert lib.rfm_fit_supported(C.byref(_cfg(max_samples=0))) == _hip.ERR_BAD_ARG
    assert lib.rfm_fit_supported(C.byref(_cfg(n_items=1))) == _hip.ERR_BAD_ARG
    assert lib.rfm_fit_supported(C.byref(_cfg(rng=_hip.RNG_MT19937))) == _hip.ERR_BAD_ARG          # MT needs serial mode
    assert lib.rfm_fit_supported(C.byref(_cfg(rng=_hip.RNG_MT19937, mode=_hip.MODE_SERIAL))) == _hip.OK
    assert lib.rfm_fit_supported(C.byref(_cfg(n_factors=513))) == _hip.ERR_UNSUPPORTED
    for F in (1, 2, 3, 10, 16, 20, 50, 64, 100, 128, 200, 256, 512):
        assert lib.rfm_fit_supported(C.byref(_cfg(n_factors=F))) == _hip.OK, F
    assert lib.rfm_fit_workspace_bytes(C.byref(_cfg(epochs=0))) == 0
    assert lib.rfm_fit_supported(None) == _hip.ERR_BAD_ARG


def test_status_strings_carry_the_reference_messages(lib):
    names = ["[w_i]", "[w_if]", "[v_u]", "[v_i]", "[v_uf]", "[v_if]"]               # order of rankfm/_rankfm.pyx:98-103
    for k, n in enumerate(names):
        msg = _hip.status_string(_hip.ERR_NONFINITE + k)
        assert n in msg and "are not finite" in msg
    assert "learning_schedule" in _hip.status_string(_hip.ERR_UNKNOWN_SCHEDULE)
    with pytest.raises(AssertionError):
        _hip.raise_for_status(_hip.ERR_NONFINITE + 2)
    with pytest.raises(ValueError):
        _hip.raise_for_status(_hip.ERR_UNKNOWN_SCHEDULE)
    with pytest.raises(_hip.EngineUnavailable):
        _hip.raise_for_status(_hip.ERR_NO_DEVICE)


def test_compute_entry_points_refuse_without_a_gpu(lib):
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    assert lib.rfm_device_count() == 0
    from rankfm_amd._rankfm import UserItemsCSR, _fit, _predict, _recommend
    X = np.array([[0, 1], [1, 2], [2, 0]], dtype=np.int32)
    csr = UserItemsCSR.from_pairs(X[:, 0], X[:, 1], 3)
    w = dict(w_i=np.zeros(3, np.float32), w_if=np.zeros(1, np.float32), v_u=np.zeros((3, 4), np.float32),
             v_i=np.zeros((3, 4), np.float32), v_uf=np.zeros((1, 4), np.float32), v_if=np.zeros((1, 4), np.float32))
    z_u, z_i = np.zeros((3, 1), np.float32), np.zeros((3, 1), np.float32)
    with pytest.raises(_hip.EngineUnavailable):
        _fit(X, np.ones(3, np.float32), csr, z_u, z_i, w["w_i"], w["w_if"], w["v_u"], w["v_i"], w["v_uf"], w["v_if"],
             0.01, 0.1, 0.1, "constant", 0.25, 1, 1, False)
    with pytest.raises(_hip.EngineUnavailable):
        _predict(np.zeros((2, 2), np.float32), z_u, z_i, w["w_i"], w["w_if"], w["v_u"], w["v_i"], w["v_uf"], w["v_if"])
    with pytest.raises(_hip.EngineUnavailable):
        _recommend(np.zeros(2, np.float32), csr, 2, False, z_u, z_i, w["w_i"], w["w_if"], w["v_u"], w["v_i"], w["v_uf"], w["v_if"])


def test_product_never_imports_the_oracle():
    """oracle/ is checker infrastructure: nothing under rankfm_amd/ may import, load or link it"""
    import glob
    for path in glob.glob(os.path.join(ROOT, "rankfm_amd", "**", "*"), recursive=True):
        if os.path.isfile(path) and path.endswith((".py", ".hip", ".hpp", ".inc", ".h")):
            text = open(path).read()
            assert "librfm_oracle" not in text and "rfm_oracle" not in text.replace("oracle/rfm_oracle.c", ""), path
            assert not re.search(r"^\s*(from|import)\s+oracle\b", text, flags=re.M), path


def test_the_library_reads_no_environment_variable():
    """the C ABI (rfm_fit_config) and nothing else steers the engine: no getenv in the sources of the library, so a stray
    environment variable cannot change the algorithm under the drop-in boundary, and the host mirror (rankfm_amd/order.py),
    which sees only the report, cannot disagree with the kernel"""
    import glob
    srcs = [p for ext in ("*.hip", "*.hpp", "*.inc") for p in glob.glob(os.path.join(ROOT, "rankfm_amd", "csrc", ext))]
    assert len(srcs) >= 4
    for p in srcs:
        assert "getenv" not in open(p).read(), p


def test_geometry_overrides_are_validated(lib):
    """the experiments' overrides live in rfm_fit_tuning behind rfm_fit_config.tuning (NULL = production): validated like the rest"""
    base = dict(n_interactions=10, n_users=4, n_items=5, n_user_features=1, n_item_features=1, n_factors=8, max_samples=1,
                epochs=1, learning_schedule=0, mode=0, rng=1)
    cfg = _hip.FitConfig(**base)
    assert not cfg.tuning and lib.rfm_fit_supported(C.byref(cfg)) == _hip.OK
    assert _hip.make_tuning() is None and _hip.make_tuning({}, debug_flags=0) is None      # (production passes a NULL pointer)
    good = _hip.make_tuning({"segment_rows": 16, "table_batch": 16}, n_workgroups=8)
    cfg.tuning = C.pointer(good)
    assert lib.rfm_fit_supported(C.byref(cfg)) == _hip.OK
    for bad in (dict(segment_rows=33), dict(hot_publications=-2), dict(table_producers=-1), dict(table_batch=6), dict(table_step_pct=500)):
        c2 = _hip.FitConfig(**base)
        c2.tuning = C.pointer(_hip.FitTuning(**bad))
        assert lib.rfm_fit_supported(C.byref(c2)) == _hip.ERR_BAD_ARG, bad
    with pytest.raises(ValueError):
        _hip.tune_kwargs({"segment_row": 3})
    # a kept engine layout lives with its plan: a layout token without a plan token is refused
    c3 = _hip.FitConfig(layout_token=7, **base)
    assert lib.rfm_fit_supported(C.byref(c3)) == _hip.ERR_BAD_ARG
    c3.plan_token = 5
    assert lib.rfm_fit_supported(C.byref(c3)) == _hip.OK


def test_the_fit_config_a_binder_fills_is_small():
    """the configuration a maintainer binds carries the reference's arguments, the engine's mode / seed and the two
    tokens of a resident caller -- the lab bench (debug flags, geometry overrides) sits behind one optional pointer"""
    names = [f for f, _ in _hip.FitConfig._fields_]
    assert len(names) <= 30 and "tuning" in names
    assert not [n for n in names if n.startswith(("debug_", "tune_"))]
