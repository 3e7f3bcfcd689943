#!/bin/bash
cd ${GRAFT_REPO_ROOT:-.}
O=gpurun_out/r05k; mkdir -p $O
( timeout 1500 python -m pytest tests/test_gpu_quality.py -x -q -m gpu -s ) > $O/quality.log 2>&1; grep -E "hit_rate|passed|failed|Error|assert" $O/quality.log | tail -30
