#!/bin/bash
cd ${GRAFT_REPO_ROOT:-.}
O=gpurun_out/r05e; mkdir -p $O
( timeout 400 python tools/ab_kernel.py --config C2 --variants "base;workgroups=160;workgroups=176;workgroups=192;workgroups=208;workgroups=224;workgroups=240;workgroups=192,hot_publications=24;workgroups=224,hot_publications=24;workgroups=208,hot_publications=32" --epochs 5 --rounds 2 ) > $O/ab_c2.log 2>&1; tail -11 $O/ab_c2.log
