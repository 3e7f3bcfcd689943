#!/bin/bash
# round 4, GPU call C: ticket ring fix; which part of the negative-side damping matters on the GPU (LL / norms from the initial weights)
cd ${GRAFT_REPO_ROOT:-.}
O=gpurun_out/r04c; mkdir -p $O
( timeout 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_edge.py -q -m gpu ) > $O/parity.log 2>&1; tail -3 $O/parity.log
( timeout 150 python tools/ab_kernel.py --config C2 --variants "base;flags=128;flags=256;flags=384" --epochs 6 --rounds 3 ) > $O/ab_c2.log 2>&1; tail -4 $O/ab_c2.log
( timeout 200 python tools/ab_kernel.py --config C2 --variants "base;flags=512;flags=1024;flags=1536;damping=64;damping=64,flags=1024;damping=32;damping=32,flags=1024" --warmup 0 --epochs 4 --rounds 1 --print-ll ) > $O/ab_c2_damp.log 2>&1; tail -16 $O/ab_c2_damp.log
( timeout 300 python tools/ab_kernel.py --config C3 --variants "base;flags=512;flags=1024;damping=64;damping=64,flags=1024;damping=32" --warmup 0 --epochs 4 --rounds 1 --print-ll ) > $O/ab_c3.log 2>&1; tail -12 $O/ab_c3.log
