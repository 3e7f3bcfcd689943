#!/bin/bash
cd ${GRAFT_REPO_ROOT:-.}
O=gpurun_out/r04i; mkdir -p $O
( timeout 1200 python tools/order_quality.py --seeds 4 --engine-variants "damping=32;damping=16;damping=8;damping=4;damping=2;damping=1;damping=0.5" ) > $O/order_quality.log 2>&1; tail -12 $O/order_quality.log
