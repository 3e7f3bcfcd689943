#!/bin/bash
cd ${GRAFT_REPO_ROOT:-.}
O=gpurun_out/r04g; mkdir -p $O
( timeout 1200 python tools/order_quality.py --seeds 4 ) > $O/order_quality.log 2>&1; tail -14 $O/order_quality.log
