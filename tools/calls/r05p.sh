#!/bin/bash
cd ${GRAFT_REPO_ROOT:-.}
O=gpurun_out/r05p; mkdir -p $O
( timeout 400 python tools/ab_kernel.py --config C4 --variants "base;flags=512;base;flags=512" --epochs 5 --rounds 3 ) > $O/ab_c4.log 2>&1; grep -v "^    " $O/ab_c4.log | tail -5 | cut -c1-200
( timeout 900 python -m pytest tests/test_gpu_configs.py -x -q -m gpu -s -k "config4" ) > $O/c4.log 2>&1; grep -E "config 4 share|passed|failed|Error|assert" $O/c4.log | cut -c1-900
