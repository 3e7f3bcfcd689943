#!/bin/bash
cd ${GRAFT_REPO_ROOT:-.}
O=gpurun_out/r05d; mkdir -p $O
( timeout 300 python tools/ab_kernel.py --config C2 --variants "base;hot_publications=12;hot_publications=24;hot_publications=96;workgroups=128;workgroups=192;flags=4" --epochs 5 --rounds 2 ) > $O/ab_c2.log 2>&1; tail -8 $O/ab_c2.log
( timeout 200 python tools/ab_kernel.py --config C2 --zipf 0 --variants "base;workgroups=512" --epochs 5 --rounds 2 ) > $O/ab_c2_uniform.log 2>&1; tail -2 $O/ab_c2_uniform.log
