"""INTEGRATION.md section 2 shows the ctypes stub a maintainer of the reference would put behind `rankfm/rankfm.py:8`.  These tests run THAT
text: the code block is cut out of the document, pointed at the in-tree library, and executed -- its struct layouts against the binding
the package itself uses (no GPU), and its `_fit` on a small problem on the GPU against the package's own drop-in `_fit`
(rankfm/_rankfm.pyx:122-142: same 19 positional arguments, in-place training, None returned)."""
import ctypes as C
import os
import re

import numpy as np
import pytest

from conftest import ROOT


def _stub_namespace():
    text = open(os.path.join(ROOT, "INTEGRATION.md")).read()
    section = text[text.index("## 2. The stub a maintainer would add"):]
    code = re.search(r"```python\n(.*?)```", section, re.S).group(1)
    assert 'C.CDLL("librankfm_hip.so")' in code
    code = code.replace('C.CDLL("librankfm_hip.so")', "C.CDLL(%r)" % os.path.join(ROOT, "rankfm_amd", "librankfm_hip.so"))
    ns = {}
    exec(compile(code, "INTEGRATION.md#stub", "exec"), ns)
    return ns


def test_the_documented_stub_has_the_layout_of_the_abi():
    import torch  # noqa: F401  (both carry a libamdhip64: the first one loaded serves the process -- INTEGRATION.md section 1)
    from rankfm_amd import _hip
    ns = _stub_namespace()
    for name, mine in (("FitConfig", _hip.FitConfig), ("FitBuffers", _hip.FitBuffers)):
        doc = ns[name]
        assert C.sizeof(doc) == C.sizeof(mine), name
        theirs = {f[0]: getattr(doc, f[0]).offset for f in doc._fields_}
        ours = {f[0]: getattr(mine, f[0]).offset for f in mine._fields_}
        assert theirs == ours, (name, set(theirs.items()) ^ set(ours.items()))
    assert ns["_lib"].rfm_abi_version() == _hip.ABI_VERSION


@pytest.mark.gpu
def test_the_documented_stub_trains_like_the_package():
    from rankfm_amd import EngineOptions, synthetic
    from rankfm_amd._rankfm import _fit
    ns = _stub_namespace()
    U, I, N, F = 800, 500, 40_000, 32
    pairs, csr = synthetic.make_interactions(U, I, N, seed=3)
    sw = np.ones(N, dtype=np.float32)
    x_uf, x_if = np.zeros((U, 1), np.float32), np.zeros((I, 1), np.float32)
    w0 = synthetic.init_weights(U, I, F, seed=7)
    user_items = {u: np.ascontiguousarray(csr.items[csr.offsets[u]:csr.offsets[u + 1]]) for u in range(U)}
    a = {k: v.copy() for k, v in w0.items()}
    b = {k: v.copy() for k, v in w0.items()}
    np.random.seed(5)
    out = ns["_fit"](pairs, sw, user_items, x_uf, x_if, a["w_i"], a["w_if"], a["v_u"], a["v_i"], a["v_uf"], a["v_if"],
                     0.01, 0.1, 0.1, "constant", 0.25, 1, 3, False)
    assert out is None
    _fit(pairs, sw, csr, x_uf, x_if, b["w_i"], b["w_if"], b["v_u"], b["v_i"], b["v_uf"], b["v_if"],
         0.01, 0.1, 0.1, "constant", 0.25, 1, 3, False, engine=EngineOptions(seed=11, device=0))
    for k in ("w_i", "v_u", "v_i"):
        assert np.isfinite(a[k]).all() and not np.array_equal(a[k], w0[k]), k
        assert abs(np.linalg.norm(a[k]) / np.linalg.norm(b[k]) - 1.0) <= 0.03, k           # (two Hogwild runs with different seeds)
    with pytest.raises(ValueError):
        ns["_fit"](pairs, sw, user_items, x_uf, x_if, a["w_i"], a["w_if"], a["v_u"], a["v_i"], a["v_uf"], a["v_if"],
                   0.01, 0.1, 0.1, "linear", 0.25, 1, 1, False)
