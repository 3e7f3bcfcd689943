#!/bin/bash
cd ${GRAFT_REPO_ROOT:-.}
O=gpurun_out/r04j; mkdir -p $O
( timeout 1200 python tools/order_quality.py --seeds 4 --engine-variants "flags=2;flags=4;flags=6;workgroups=192" ) > $O/order_quality.log 2>&1; tail -8 $O/order_quality.log
