#!/bin/bash
# A/B of two BUILDS on one box: rankfm_amd/librankfm_hip_prev.so (the committed tree's library, copied aside) against the current one,
# alternating processes.   bash tools/calls/ab_builds.sh <out-dir> <ab_kernel args...>
cd ${GRAFT_REPO_ROOT:-.}
O=gpurun_out/$1; shift; mkdir -p $O
cp rankfm_amd/librankfm_hip.so /tmp/lib_new.so
cp rankfm_amd/librankfm_hip_prev.so /tmp/lib_prev.so
for rep in $(seq 1 ${AB_REPS:-2}); do
  for which in prev new; do
    cp /tmp/lib_$which.so rankfm_amd/librankfm_hip.so
    echo "== $which (repetition $rep)" >> $O/ab.log
    timeout 300 python tools/ab_kernel.py "$@" 2>&1 | grep -v "amdgpu.ids" >> $O/ab.log
  done
done
cp /tmp/lib_new.so rankfm_amd/librankfm_hip.so
grep -E "^==|kernel ms" $O/ab.log
