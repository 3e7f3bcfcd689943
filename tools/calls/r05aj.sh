#!/bin/bash
cd ${GRAFT_REPO_ROOT:-.}
O=gpurun_out/r05aj; mkdir -p $O
( timeout 600 python tools/ab_kernel.py --config C3 --variants "base;workgroups=224;workgroups=192;workgroups=160;base" --epochs 5 --rounds 2 2>&1 | grep -v amdgpu.ids ) > $O/c3_wg.log 2>&1; cat $O/c3_wg.log
( timeout 600 python tools/ab_kernel.py --config C5 --variants "base;workgroups=224;workgroups=192" --epochs 2 --rounds 2 --warmup 1 2>&1 | grep -v amdgpu.ids ) > $O/c5_wg.log 2>&1; cat $O/c5_wg.log
