#!/bin/bash
cd ${GRAFT_REPO_ROOT:-.}
O=gpurun_out/r05u; mkdir -p $O
( timeout 900 python -m pytest tests/test_gpu_configs.py -x -q -m gpu -s -k "config4" ) > $O/c4.log 2>&1; grep -E "config 4 share|passed|failed|Error|assert" $O/c4.log | cut -c1-700
for v in "table_every=223" "table_every=150"; do
  echo "== tune: ${v:-default}" >> $O/tags_sweep.log
  ( timeout 400 python tools/calls/tags_runs.py 2 $v ) >> $O/tags_sweep.log 2>&1
done
grep -E "^==|^seed" $O/tags_sweep.log | cut -c1-200
( timeout 2400 python -m pytest tests -x -q -m gpu --deselect tests/test_gpu_configs.py::test_config4_one_gpu_share_with_features_tracks_sequential_oracle ) > $O/gpu_suite_rest.log 2>&1; tail -5 $O/gpu_suite_rest.log
