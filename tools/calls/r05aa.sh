#!/bin/bash
cd ${GRAFT_REPO_ROOT:-.}
O=gpurun_out/r05aa; mkdir -p $O
( timeout 1200 python -m pytest tests/test_gpu_quality.py -x -q -m gpu -s -k "asynchrony or bpr_k32 or bpr_k64" ) > $O/async.log 2>&1; grep -E "asynchrony by itself|config-2 shape bpr|passed|failed" $O/async.log | cut -c1-500
