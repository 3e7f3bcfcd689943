#!/bin/bash
cd ${GRAFT_REPO_ROOT:-.}
O=gpurun_out/r04o; mkdir -p $O
( timeout 300 python -m pytest tests/test_gpu_parity.py -q -m gpu -x ) > $O/parity.log 2>&1; tail -2 $O/parity.log
( timeout 200 python tools/ab_kernel.py --config C2 --variants "base;sweep_every=2;sweep_every=4;hot_publications=24;hot_publications=24,sweep_every=2;hot_publications=12,sweep_every=4" --warmup 0 --epochs 4 --rounds 1 --print-ll ) > $O/ab_c2_ll.log 2>&1; tail -12 $O/ab_c2_ll.log
( timeout 200 python tools/ab_kernel.py --config C2 --variants "base;sweep_every=2;sweep_every=4;hot_publications=24;hot_publications=24,sweep_every=2;hot_publications=12,sweep_every=4" --epochs 6 --rounds 3 ) > $O/ab_c2.log 2>&1; tail -6 $O/ab_c2.log
