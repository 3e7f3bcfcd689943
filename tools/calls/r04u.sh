#!/bin/bash
cd ${GRAFT_REPO_ROOT:-.}
O=gpurun_out/r04u; mkdir -p $O
( timeout 900 python tools/merge_engine_scan.py --world 8 --epochs 5 --seeds 2 --variants "0.1:0.3:auto;0.1:0.3:1;0.1:0.3:8" ) > $O/scan_e5.log 2>&1; tail -6 $O/scan_e5.log
( timeout 900 python tools/merge_engine_scan.py --world 8 --epochs 15 --seeds 2 --variants "0.1:0.3:auto;0.1:0.3:1" ) > $O/scan_e15.log 2>&1; tail -5 $O/scan_e15.log
( timeout 900 python tools/merge_engine_scan.py --world 8 --epochs 2 --seeds 2 --variants "0.1:0.3:auto;0.1:0.3:1" ) > $O/scan_e2.log 2>&1; tail -5 $O/scan_e2.log
