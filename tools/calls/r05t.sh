#!/bin/bash
cd ${GRAFT_REPO_ROOT:-.}
O=$(pwd)/gpurun_out/r05t; mkdir -p $O
cd /tmp && export TMPDIR=/tmp && cd ${GRAFT_REPO_ROOT:-.}
( timeout 600 rocprofv3 --kernel-trace --stats -d $O/infer -o out --output-format csv -- python tools/infer_timing.py ) > $O/infer_rocprof.log 2>&1; tail -3 $O/infer_rocprof.log
cp $(find $O/infer -name "*kernel_stats.csv" | head -1) $O/r05_infer_kernel_stats.csv; head -12 $O/r05_infer_kernel_stats.csv | cut -c1-200
