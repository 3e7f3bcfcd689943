#!/bin/bash
cd ${GRAFT_REPO_ROOT:-.}
O=gpurun_out/r05ai; mkdir -p $O
( timeout 600 python -m pytest tests/test_gpu_edge.py -q -m gpu -x -k "independent_rows" 2>&1 | tail -25 ) > $O/indep.log 2>&1; cat $O/indep.log
