#!/bin/bash
cd ${GRAFT_REPO_ROOT:-.}
O=gpurun_out/r05ag; mkdir -p $O
( timeout 900 python -m pytest tests/test_gpu_configs.py -q -m gpu -k "config3 or config5" -x 2>&1 | tail -5 ) > $O/c3_parity.log 2>&1; cat $O/c3_parity.log
( timeout 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_edge.py -q -m gpu -x 2>&1 | tail -3 ) > $O/warp_parity.log 2>&1; cat $O/warp_parity.log
AB_REPS=2 bash tools/calls/ab_builds.sh r05ag --config C3 --variants base --epochs 6 --rounds 2
AB_REPS=1 bash tools/calls/ab_builds.sh r05ag_c5 --config C5 --variants base --epochs 2 --rounds 2
