#!/usr/bin/env bash
# Build the REFERENCE's own `_fit` (rankfm/_rankfm.pyx + its vendored mt19937ar.c) into
# oracle/_ref/ -- THIS CONTAINER ONLY.  Test infrastructure, not product code.
#
# * Sources are compiled where they lie under /root/reference; nothing from the reference
#   is copied into the repository.  The only outputs are the Cython-generated C file and
#   the extension module, both inside oracle/_ref/ (git-ignored AND gpurun-ignored: the
#   reference is a Python package, so it must not travel to the GPU box in any form).
# * The checked-in rankfm/_rankfm.c (Cython 0.29.2 output) does not compile on Python 3.10
#   (tp_print & friends), so the C is regenerated from the .pyx with the image's Cython;
#   compile flags are the reference's own (setup.py:25-26: -O2 -ffast-math).
# * Used by tests/golden/make_golden.py (golden vectors) and tools/calibrate_cpu.py
#   (restatement-vs-reference timing ratio).  Never imported by the product or by bench.py.
set -euo pipefail
REF=${RANKFM_REFERENCE:-/root/reference}
HERE="$(cd "$(dirname "${BASH_SOURCE[0]}")" && pwd)"
OUT="$HERE/_ref"
[ -f "$REF/rankfm/_rankfm.pyx" ] || { echo "reference not present at $REF - skipping"; exit 0; }
mkdir -p "$OUT"
PYINC=$(python3 -c "import sysconfig; print(sysconfig.get_paths()['include'])")
NPINC=$(python3 -c "import numpy; print(numpy.get_include())")
EXT=$(python3 -c "import sysconfig; print(sysconfig.get_config_var('EXT_SUFFIX'))")
cython -3 -I "$REF" "$REF/rankfm/_rankfm.pyx" -o "$OUT/_rankfm.c"
gcc -shared -fPIC -O2 -ffast-math -Wno-unused-function -Wno-uninitialized \
    -I "$PYINC" -I "$NPINC" -I "$REF/rankfm" \
    "$OUT/_rankfm.c" "$REF/rankfm/mt19937ar/mt19937ar.c" -o "$OUT/_rankfm$EXT"
echo "built $OUT/_rankfm$EXT"
