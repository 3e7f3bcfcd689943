#!/bin/bash
cd ${GRAFT_REPO_ROOT:-.}
O=gpurun_out/r05am; mkdir -p $O
( timeout 600 python -m pytest tests/test_gpu_api.py -q -m gpu -x -k "resident or recommend or predict or similar or save_load" 2>&1 | tail -8 ) > $O/api.log 2>&1; cat $O/api.log
( timeout 300 python tools/infer_timing.py 2>&1 | grep -v amdgpu.ids ) > $O/infer.log 2>&1; cat $O/infer.log
