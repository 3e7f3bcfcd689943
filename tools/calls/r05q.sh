#!/bin/bash
cd ${GRAFT_REPO_ROOT:-.}
O=gpurun_out/r05q; mkdir -p $O
for v in "" "table_every=223" "table_every=300" "table_every=650" "table_every=892"; do
  echo "== tune: ${v:-default}" >> $O/tags_sweep.log
  ( timeout 400 python tools/calls/tags_runs.py 2 $v ) >> $O/tags_sweep.log 2>&1
done
grep -E "^==|^seed" $O/tags_sweep.log | cut -c1-330
