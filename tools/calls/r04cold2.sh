#!/bin/bash
cd ${GRAFT_REPO_ROOT:-.}
bash tools/calls/ab_builds.sh r04cold_c4 --config C4 --variants "base" --epochs 4 --rounds 2 --warmup 3
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_configs.py tests/test_gpu_edge.py -x -q -m gpu 2>&1 | tail -3
