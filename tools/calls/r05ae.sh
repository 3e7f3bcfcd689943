#!/bin/bash
cd ${GRAFT_REPO_ROOT:-.}
O=gpurun_out/${OUTDIR:-r05ae}; mkdir -p $O
cp rankfm_amd/librankfm_hip.so /tmp/lib_new.so
cp rankfm_amd/librankfm_hip_prev.so /tmp/lib_prev.so
for C in ${CONFIGS:-C3 C5}; do
cp /tmp/lib_prev.so rankfm_amd/librankfm_hip.so
timeout 600 python tools/frozen_epoch_timing.py --config $C --train 6 --save /tmp/w_$C.npz 2>&1 | grep -v amdgpu.ids >> $O/frozen.log
for which in prev new prev new; do
  cp /tmp/lib_$which.so rankfm_amd/librankfm_hip.so
  echo "== $C $which" >> $O/frozen.log
  timeout 600 python tools/frozen_epoch_timing.py --config $C --train 6 --load /tmp/w_$C.npz --reps 3 2>&1 | grep -v amdgpu.ids >> $O/frozen.log
done
done
cp /tmp/lib_new.so rankfm_amd/librankfm_hip.so
cat $O/frozen.log
