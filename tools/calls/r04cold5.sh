#!/bin/bash
cd ${GRAFT_REPO_ROOT:-.}
AB_REPS=4 bash tools/calls/ab_builds.sh r04cold5_c2 --config C2 --variants "base" --epochs 6 --rounds 3 --warmup 3
O=gpurun_out/r04cold5_c2
for r in 1 2; do
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -s -m gpu -k "config2_tracks" > $O/pytest$r.log 2>&1; grep -E "full-size|passed|failed" $O/pytest$r.log | cut -c1-330
done
