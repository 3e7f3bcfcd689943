#!/bin/bash
cd ${GRAFT_REPO_ROOT:-.}
O=gpurun_out/r04feat; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_quality.py tests/test_gpu_configs.py -q -s -m gpu -k "feature or tags or c4 or C4 or config4" > $O/pytest.log 2>&1; grep -E "config-2 shape|passed|failed|hit_rate|FAILED|Error" $O/pytest.log | cut -c1-400 | tail -12
