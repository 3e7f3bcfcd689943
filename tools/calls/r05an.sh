#!/bin/bash
cd ${GRAFT_REPO_ROOT:-.}
O=gpurun_out/r05an; mkdir -p $O
( timeout 900 python -m pytest tests/test_gpu_api.py tests/test_gpu_edge.py -q -m gpu -x 2>&1 | tail -4 ) > $O/api.log 2>&1; cat $O/api.log
( timeout 600 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | grep -v amdgpu.ids ) > $O/smoke.log 2>&1; tail -4 $O/smoke.log
