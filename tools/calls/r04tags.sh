#!/bin/bash
cd ${GRAFT_REPO_ROOT:-.}
O=gpurun_out/r04tags; mkdir -p $O
cp rankfm_amd/librankfm_hip.so /tmp/lib_new.so; cp rankfm_amd/librankfm_hip_prev.so /tmp/lib_prev.so
for which in prev new; do
  cp /tmp/lib_$which.so rankfm_amd/librankfm_hip.so
  echo "== $which" >> $O/tags.log
  timeout 600 python tools/calls/tags_runs.py 3 2>&1 | grep -v amdgpu.ids >> $O/tags.log
done
cp /tmp/lib_new.so rankfm_amd/librankfm_hip.so
cat $O/tags.log
