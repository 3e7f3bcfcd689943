#!/bin/bash
cd ${GRAFT_REPO_ROOT:-.}
O=gpurun_out/r04z; mkdir -p $O
( timeout 1500 python -m pytest tests/test_gpu_quality.py -q -m gpu -s -k "config2_shape or eight_engine" ) > $O/quality_c2.log 2>&1; grep -E "config-2 shape|eight engine|passed|failed|Error" $O/quality_c2.log | cut -c1-500
