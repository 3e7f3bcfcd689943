#!/bin/bash
cd ${GRAFT_REPO_ROOT:-.}
O=gpurun_out/r05r; mkdir -p $O
for v in "" "table_every=123" "table_every=491"; do
  echo "== tune: ${v:-default}" >> $O/tags_sweep.log
  ( timeout 400 python tools/calls/tags_runs.py 2 $v ) >> $O/tags_sweep.log 2>&1
done
grep -E "^==|^seed" $O/tags_sweep.log | cut -c1-200
( FQ_RUNS=2 timeout 600 python tools/feature_quality.py "" "table_every=5" "table_every=10" "table_every=20" "table_every=40" ) > $O/fixture.log 2>&1; tail -7 $O/fixture.log | cut -c1-330
( timeout 900 python -m pytest tests/test_gpu_configs.py -x -q -m gpu -s -k "config4" ) > $O/c4.log 2>&1; grep -E "config 4 share|passed|failed|Error|assert" $O/c4.log | cut -c1-900
