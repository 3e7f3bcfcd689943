#!/bin/bash
cd ${GRAFT_REPO_ROOT:-.}
O=gpurun_out/r04full2; mkdir -p $O
( timeout 300 python tools/ab_kernel.py --config C2 --variants "base;flags=1024;base;flags=1024;base;flags=1024" --epochs 6 --rounds 3 ) > $O/ab_c2.log 2>&1; tail -6 $O/ab_c2.log
( timeout 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_edge.py -q -m gpu -x ) > $O/parity.log 2>&1; tail -2 $O/parity.log
