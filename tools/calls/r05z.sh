#!/bin/bash
cd ${GRAFT_REPO_ROOT:-.}
O=gpurun_out/r05z; mkdir -p $O
( timeout 400 python tools/ab_kernel.py --config C4 --variants "base;hot_publications=32;hot_publications=24;base;hot_publications=32;hot_publications=24" --epochs 5 --rounds 3 ) > $O/ab_c4.log 2>&1; grep -v "^    " $O/ab_c4.log | tail -6 | cut -c1-200
( timeout 400 python tools/ab_kernel.py --config C3 --variants "base;hot_publications=32;base;hot_publications=32" --epochs 5 --rounds 2 ) > $O/ab_c3.log 2>&1; tail -4 $O/ab_c3.log | cut -c1-200
