#!/bin/bash
cd ${GRAFT_REPO_ROOT:-.}
O=gpurun_out/r04k; mkdir -p $O
( timeout 300 python tools/ab_kernel.py --config C4 --variants "base;feature_waves=16;feature_waves=8;table_every=160;table_every=640;flags=128" --epochs 4 --rounds 2 ) > $O/ab_c4.log 2>&1; tail -14 $O/ab_c4.log
( timeout 900 python -m pytest tests -q -m gpu -x -k "feature or config4 or tags" -s ) > $O/feat_tests.log 2>&1; grep -E "passed|failed|Error|feature model|config 4" $O/feat_tests.log | cut -c1-400 | tail -12
