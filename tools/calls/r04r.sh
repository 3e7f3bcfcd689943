#!/bin/bash
cd ${GRAFT_REPO_ROOT:-.}
O=gpurun_out/r04r; mkdir -p $O
( timeout 900 python tools/merge_engine_scan.py --variants "0.1:0.3:1;0.03:0.3:1;0.01:0.3:1;0.003:0.3:1;0.0003:0.3:1;0.01:0.1:1;0.01:1.0:1;0.003:0.3:2" ) > $O/scan.log 2>&1; tail -12 $O/scan.log
