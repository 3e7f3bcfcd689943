#!/bin/bash
cd ${GRAFT_REPO_ROOT:-.}
O=gpurun_out/r05ak; mkdir -p $O
( timeout 600 python tools/ab_kernel.py --config C3 --variants "base;flags=1024;base;flags=1024" --epochs 6 --rounds 2 2>&1 | grep -v amdgpu.ids ) > $O/ab_c3_wide.log 2>&1; cat $O/ab_c3_wide.log
