"""Build librankfm_hip.so (gfx950 only) in-tree with hipcc.  Cross-compiles without a GPU.

    python -m rankfm_amd._build [--force]

One translation unit per SGD row-group shape (compiled in parallel) + the C-ABI host code.
The .so is git-ignored but travels with the repository snapshot to the GPU box.
"""
import concurrent.futures
import glob
import os
import shutil
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
OBJ = os.path.join(CSRC, "_obj")
LIB = os.path.join(HERE, "librankfm_hip.so")
ARCH = "gfx950"
FLAGS = ["--offload-arch=" + ARCH, "-O3", "-std=c++17", "-fPIC", "-munsafe-fp-atomics", "-ffp-contract=off",
         "-Wall", "-Wno-unused-function",
         # LDS / memory atomics of a few lanes on one address are cheaper left to the hardware than turned into a scalar loop
         "-mllvm", "-amdgpu-atomic-optimizer-strategy=None"]


def _hipcc():
    exe = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(exe):
        raise RuntimeError("hipcc not found: librankfm_hip.so cannot be built (and there is no CPU fallback)")
    return exe


def _deps():
    return (glob.glob(os.path.join(CSRC, "*.hpp")) + glob.glob(os.path.join(CSRC, "*.inc"))
            + glob.glob(os.path.join(HERE, "..", "include", "*.h")))


def _stale(target, sources):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(s) > t for s in sources)


def _compile(src):
    obj = os.path.join(OBJ, os.path.basename(src)[:-4] + ".o")
    # RFM_BUILD_SHAPES="g16_k4 ..." (kernel development only): recompile just these row-group shapes and the host code and link
    # the other shapes' existing objects.  Only valid while SgdArgs is unchanged; a normal build recompiles everything stale.
    only = os.environ.get("RFM_BUILD_SHAPES", "").split()
    if only and os.path.exists(obj) and "rfm_sgd_inst_" in src and not any(t in src for t in only):
        return obj
    if _stale(obj, [src] + _deps()):
        subprocess.check_call([_hipcc()] + FLAGS + ["-c", src, "-o", obj])
    return obj


def build(force=False, verbose=False):
    srcs = sorted(glob.glob(os.path.join(CSRC, "*.hip")))
    if force and os.path.isdir(OBJ):
        shutil.rmtree(OBJ)
    os.makedirs(OBJ, exist_ok=True)
    # every object is checked against its source and the shared headers (a partial RFM_BUILD_SHAPES build leaves the library newer
    # than objects that are themselves out of date), then the library against the objects
    with concurrent.futures.ThreadPoolExecutor(max_workers=min(8, os.cpu_count() or 1)) as ex:
        objs = list(ex.map(_compile, srcs))
    if not force and not _stale(LIB, objs):
        return LIB
    subprocess.check_call([_hipcc(), "--offload-arch=" + ARCH, "-shared", "-fPIC", "-o", LIB] + objs)
    if verbose:
        print("built", LIB)
    return LIB


if __name__ == "__main__":
    build(force="--force" in sys.argv, verbose=True)
