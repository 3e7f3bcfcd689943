#!/bin/bash
cd ${GRAFT_REPO_ROOT:-.}
O=gpurun_out/r05b; mkdir -p $O
( timeout 120 tools/microbench/atomic_skew ) > $O/atomic_skew.log 2>&1; cat $O/atomic_skew.log
