#!/bin/bash
cd ${GRAFT_REPO_ROOT:-.}
O=gpurun_out/r05c; mkdir -p $O
( timeout 120 tools/microbench/pipe_model2 ) > $O/pipe_model2.log 2>&1; cat $O/pipe_model2.log
