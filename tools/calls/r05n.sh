#!/bin/bash
cd ${GRAFT_REPO_ROOT:-.}
O=gpurun_out/r05n; mkdir -p $O
( timeout 400 python tools/ab_kernel.py --config C2 --variants "base;workgroups=192;workgroups=224;base;workgroups=192;workgroups=224;base;workgroups=192;workgroups=224" --epochs 5 --rounds 3 ) > $O/ab_c2_wg.log 2>&1; tail -9 $O/ab_c2_wg.log | cut -c1-120
( timeout 1500 python -m pytest tests/test_gpu_quality.py -x -q -m gpu -s -k "eight_engine or asynchrony" ) > $O/quality2.log 2>&1; grep -E "eight engine|asynchrony|passed|failed|Error" $O/quality2.log | cut -c1-700
