#!/bin/bash
cd ${GRAFT_REPO_ROOT:-.}
O=gpurun_out/r05x; mkdir -p $O
( timeout 300 python tools/c1_timing.py 2 ) > $O/c1_timing.log 2>&1; tail -6 $O/c1_timing.log | cut -c1-200
( timeout 2400 python -m pytest tests/test_gpu_quality.py tests/test_gpu_configs.py -x -q -m gpu -s ) > $O/quality_configs.log 2>&1; grep -E "table quota sweep|config 4 share|passed|failed|Error" $O/quality_configs.log | cut -c1-330
