#!/bin/bash
cd ${GRAFT_REPO_ROOT:-.}
O=gpurun_out/r04s; mkdir -p $O
( timeout 900 python tools/merge_engine_scan.py --world 1 --variants "0.1:0.3:1" ) > $O/scan_w1.log 2>&1; tail -4 $O/scan_w1.log
( timeout 900 python tools/merge_engine_scan.py --world 2 --variants "0.1:0.3:1;0.03:0.3:1" ) > $O/scan_w2.log 2>&1; tail -5 $O/scan_w2.log
( timeout 900 python tools/merge_engine_scan.py --world 8 --variants "0.1:0.3:8;0.1:0.3:32;0.03:0.3:8;0.03:0.3:32" ) > $O/scan_w8.log 2>&1; tail -7 $O/scan_w8.log
