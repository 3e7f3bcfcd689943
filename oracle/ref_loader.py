"""Load the REFERENCE rankfm package for golden-vector generation -- THIS CONTAINER ONLY.

Test infrastructure.  Nothing in `rankfm_amd/` or `bench.py` may import this file.

The reference's Python sources are read where they lie (/root/reference, read-only); the
compiled `_fit/_predict/_recommend` extension comes from `oracle/_ref/` (built by
`oracle/build_ref.sh`).  The extension is registered as `rankfm._rankfm` before the
package is imported, so `rankfm/rankfm.py:8` (`from rankfm._rankfm import ...`) resolves
to it.  Neither the sources nor the extension travel to the GPU box (.gpurunignore).
"""
import glob
import importlib.machinery
import importlib.util
import os
import sys

REFERENCE_ROOT = os.environ.get("RANKFM_REFERENCE", "/root/reference")
_REF_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "_ref")


def available():
    return bool(glob.glob(os.path.join(_REF_DIR, "_rankfm*.so"))) and os.path.isdir(
        os.path.join(REFERENCE_ROOT, "rankfm"))


def load_reference():
    """Returns (RankFM class, _rankfm extension module, evaluation module) of the reference."""
    if "rankfm._rankfm" not in sys.modules:
        so = sorted(glob.glob(os.path.join(_REF_DIR, "_rankfm*.so")))
        if not so:
            raise RuntimeError("oracle/_ref not built: run oracle/build_ref.sh (needs /root/reference)")
        loader = importlib.machinery.ExtensionFileLoader("rankfm._rankfm", so[0])
        spec = importlib.util.spec_from_file_location("rankfm._rankfm", so[0], loader=loader)
        mod = importlib.util.module_from_spec(spec)
        sys.modules["rankfm._rankfm"] = mod
        loader.exec_module(mod)
    if REFERENCE_ROOT not in sys.path:
        sys.path.insert(0, REFERENCE_ROOT)
    from rankfm.rankfm import RankFM          # noqa: E402  (reference code, imported not copied)
    import rankfm.evaluation as evaluation    # noqa: E402
    return RankFM, sys.modules["rankfm._rankfm"], evaluation
