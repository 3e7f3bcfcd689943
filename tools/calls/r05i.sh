#!/bin/bash
cd ${GRAFT_REPO_ROOT:-.}
O=gpurun_out/r05i; mkdir -p $O
( timeout 600 python tools/pub_margin.py --pubs 48,32,24,16 --runs 2 ) > $O/pub_margin.log 2>&1; tail -9 $O/pub_margin.log
