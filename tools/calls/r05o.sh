#!/bin/bash
cd ${GRAFT_REPO_ROOT:-.}
O=gpurun_out/r05o; mkdir -p $O
( timeout 1500 python tools/table_step_scan.py --steps 100,70,50 --every 0,223,892 ) > $O/table_step_scan.log 2>&1; grep "^step" $O/table_step_scan.log | cut -c1-260; tail -3 $O/table_step_scan.log | cut -c1-300
