#!/bin/bash
cd ${GRAFT_REPO_ROOT:-.}
O=gpurun_out/r05s; mkdir -p $O
( timeout 600 python -m pytest tests/test_gpu_api.py tests/test_gpu_edge.py -x -q -m gpu ) > $O/api.log 2>&1; tail -4 $O/api.log
( timeout 300 python tools/infer_timing.py ) > $O/infer.log 2>&1; tail -3 $O/infer.log
( timeout 1500 python -m pytest tests/test_gpu_quality.py -x -q -m gpu -s -k "table_quota_sweep" ) > $O/sweep.log 2>&1; grep -E "table quota sweep|passed|failed|Error|assert" $O/sweep.log | cut -c1-400
