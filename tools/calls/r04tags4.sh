#!/bin/bash
cd ${GRAFT_REPO_ROOT:-.}
O=gpurun_out/r04tags4; mkdir -p $O
run() { echo "== $*" >> $O/tags.log; timeout 900 python tools/calls/tags_runs.py 1 users:125000 items:200000 tags:16 factors:64 lr:0.03 epochs:10 seeds:3 "$@" 2>&1 | grep -v amdgpu.ids | cut -c1-300 >> $O/tags.log; }
run
run table_every=450
run table_every=280
cat $O/tags.log
