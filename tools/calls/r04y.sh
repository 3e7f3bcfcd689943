#!/bin/bash
cd ${GRAFT_REPO_ROOT:-.}
O=gpurun_out/r04y; mkdir -p $O
( timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_edge.py -q -m gpu -x ) > $O/parity.log 2>&1; tail -3 $O/parity.log
( timeout 300 python tools/ab_kernel.py --config C3 --variants "base;flags=512;base;flags=512" --warmup 3 --epochs 4 --rounds 2 ) > $O/ab_c3.log 2>&1; tail -4 $O/ab_c3.log
( timeout 900 python -m pytest tests/test_gpu_configs.py -q -m gpu -x -s -k "config3 or config5" ) > $O/configs.log 2>&1; grep -E "passed|failed|config 3 vs|config 5|Error" $O/configs.log | cut -c1-300
