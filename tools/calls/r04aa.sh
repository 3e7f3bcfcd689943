#!/bin/bash
cd ${GRAFT_REPO_ROOT:-.}
O=gpurun_out/r04aa; mkdir -p $O
( timeout 600 python tools/host_path_timing.py ) > $O/host_path.log 2>&1; grep -v amdgpu.ids $O/host_path.log | tail -9
( timeout 900 python -m pytest tests/test_gpu_api.py tests/test_gpu_edge.py -q -m gpu -x ) > $O/api_tests.log 2>&1; tail -2 $O/api_tests.log
