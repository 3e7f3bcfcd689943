#!/bin/bash
cd ${GRAFT_REPO_ROOT:-.}
O=gpurun_out/r04last; mkdir -p $O
timeout 150 python -m pytest tests/test_gpu_parity.py -q -s -m gpu -k "config2_tracks" > $O/pytest.log 2>&1; grep -E "full-size|passed|failed" $O/pytest.log | cut -c1-330
