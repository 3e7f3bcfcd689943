#!/bin/bash
cd ${GRAFT_REPO_ROOT:-.}
bash tools/calls/ab_builds.sh r04cold_c3 --config C3 --variants "base" --epochs 4 --rounds 2 --warmup 6
bash tools/calls/ab_builds.sh r04cold_c5 --config C5 --variants "base" --epochs 2 --rounds 1 --warmup 3
timeout 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_configs.py -x -q -m gpu -k "warp or WARP or c3 or C3 or c5 or C5" 2>&1 | tail -3
