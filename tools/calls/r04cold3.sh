#!/bin/bash
cd ${GRAFT_REPO_ROOT:-.}
bash tools/calls/ab_builds.sh r04cold_c2 --config C2 --variants "base" --epochs 6 --rounds 3 --warmup 3
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_edge.py -x -q -m gpu 2>&1 | tail -3
