#!/bin/bash
cd ${GRAFT_REPO_ROOT:-.}
O=gpurun_out/r04q; mkdir -p $O
( timeout 900 python -m pytest tests/test_gpu_distributed.py -q -m gpu -x ) > $O/dist_tests.log 2>&1; tail -3 $O/dist_tests.log
( timeout 1500 python -m pytest tests/test_gpu_quality.py -q -m gpu -s -k "eight_engine" ) > $O/emul.log 2>&1; grep -E "eight engine shards|passed|failed|Error" $O/emul.log | cut -c1-400
