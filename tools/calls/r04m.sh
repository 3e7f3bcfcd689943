#!/bin/bash
cd ${GRAFT_REPO_ROOT:-.}
O=gpurun_out/r04m; mkdir -p $O
( timeout 300 python tools/ab_kernel.py --config C4 --variants "base;table_every=500;base" --epochs 4 --rounds 2 ) > $O/ab_c4.log 2>&1; tail -6 $O/ab_c4.log
for k in 1 2 3; do ( timeout 600 python -m pytest tests/test_gpu_configs.py -q -m gpu -x -k "config4" -s ) > $O/c4_test_$k.log 2>&1; grep -E "passed|failed|config 4 share" $O/c4_test_$k.log | cut -c1-520; done
( timeout 1500 python -m pytest tests -q -m gpu -x ) > $O/gputests.log 2>&1; tail -4 $O/gputests.log
