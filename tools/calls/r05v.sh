#!/bin/bash
cd ${GRAFT_REPO_ROOT:-.}
AB_REPS=2 bash tools/calls/ab_builds.sh r05v_c3 --config C3 --variants "base" --epochs 5 --rounds 2 | cut -c1-150
AB_REPS=2 bash tools/calls/ab_builds.sh r05v_c5 --config C5 --variants "base" --epochs 3 --rounds 2 --warmup 2 | cut -c1-150
( timeout 900 python -m pytest tests/test_gpu_configs.py tests/test_gpu_parity.py -x -q -m gpu -k "config3 or config5 or warp" ) > gpurun_out/r05v_tests.log 2>&1; tail -3 gpurun_out/r05v_tests.log
