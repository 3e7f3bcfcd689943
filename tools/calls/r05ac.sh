#!/bin/bash
cd ${GRAFT_REPO_ROOT:-.}
O=gpurun_out/r05ac; mkdir -p $O
( timeout 600 python tools/pub_margin.py --pubs 32 --runs 3 ) > $O/pub_margin_final.log 2>&1; tail -3 $O/pub_margin_final.log
( timeout 300 python tools/infer_timing.py ) > $O/infer.log 2>&1; tail -2 $O/infer.log
