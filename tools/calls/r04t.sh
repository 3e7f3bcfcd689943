#!/bin/bash
cd ${GRAFT_REPO_ROOT:-.}
O=gpurun_out/r04t; mkdir -p $O
( timeout 900 python tools/merge_engine_scan.py --world 8 --epochs 15 --variants "0.1:0.3:1;0.03:0.3:1;0.1:0.3:2;0.1:0.3:4;0.1:0.3:8" ) > $O/scan_e15.log 2>&1; tail -8 $O/scan_e15.log
( timeout 900 python tools/merge_engine_scan.py --world 8 --epochs 40 --variants "0.1:0.3:1;0.1:0.3:4" ) > $O/scan_e40.log 2>&1; tail -5 $O/scan_e40.log
