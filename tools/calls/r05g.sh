#!/bin/bash
cd ${GRAFT_REPO_ROOT:-.}
O=gpurun_out/r05g; mkdir -p $O
( timeout 400 python tools/ab_kernel.py --config C2 --variants "base;flags=512;base;flags=512;hot_publications=24;flags=512,hot_publications=24" --epochs 5 --rounds 3 --print-ll ) > $O/ab_c2.log 2>&1; grep -v "LL per epoch" $O/ab_c2.log | tail -7
( timeout 600 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "c2 or full or hogwild or segments" ) > $O/parity.log 2>&1; tail -5 $O/parity.log
