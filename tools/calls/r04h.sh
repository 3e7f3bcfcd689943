#!/bin/bash
cd ${GRAFT_REPO_ROOT:-.}
O=gpurun_out/r04h; mkdir -p $O
( timeout 1200 python tools/order_quality.py --seeds 4 --engine-variants "hot_publications=96;hot_publications=192;hot_publications=384;damping=64;damping=256;damping=64,hot_publications=192;flags=128;flags=256" ) > $O/order_quality.log 2>&1; tail -14 $O/order_quality.log
