#!/bin/bash
cd ${GRAFT_REPO_ROOT:-.}
O=gpurun_out/r04e; mkdir -p $O
( timeout 200 python tools/ab_kernel.py --config C2 --variants "base;flags=256;damping=64;damping=32;damping=256;flags=128" --warmup 0 --epochs 4 --rounds 1 --print-ll ) > $O/ab_c2_damp.log 2>&1; tail -12 $O/ab_c2_damp.log
( timeout 300 python tools/ab_kernel.py --config C3 --variants "base;flags=256;damping=64;damping=32;flags=128" --warmup 0 --epochs 4 --rounds 1 --print-ll ) > $O/ab_c3.log 2>&1; tail -10 $O/ab_c3.log
( timeout 900 python -m pytest tests -q -m gpu -x --deselect tests/test_gpu_quality.py ) > $O/gputests.log 2>&1; tail -5 $O/gputests.log
