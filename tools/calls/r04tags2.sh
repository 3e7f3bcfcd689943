#!/bin/bash
cd ${GRAFT_REPO_ROOT:-.}
O=gpurun_out/r04tags2; mkdir -p $O
cp rankfm_amd/librankfm_hip.so /tmp/lib_new.so; cp rankfm_amd/librankfm_hip_prev.so /tmp/lib_prev.so
run() { cp /tmp/lib_$1.so rankfm_amd/librankfm_hip.so; shift; echo "== $*" >> $O/tags.log; timeout 600 python tools/calls/tags_runs.py 2 "$@" 2>&1 | grep -v amdgpu.ids | cut -c1-260 >> $O/tags.log; }
run new table_every=250
run new table_every=180
run prev table_every=450
run prev table_every=250
run new table_producers=2
cp /tmp/lib_new.so rankfm_amd/librankfm_hip.so
cat $O/tags.log
