#!/bin/bash
cd ${GRAFT_REPO_ROOT:-.}
O=gpurun_out/r04cold4; mkdir -p $O
for r in 1 2 3 4; do
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -s -m gpu -k "config2_tracks and stripes" > $O/pytest$r.log 2>&1; grep -E "full-size|passed|failed|Error|assert" $O/pytest$r.log | cut -c1-400
done
