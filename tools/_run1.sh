#!/bin/bash
# banking run: new config tests + kernel statistics of configs 3/4/5 with the round-1 kernels
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export RFM_DATA_CACHE=/tmp/rfmc
( time timeout 1200 python -m pytest tests/test_gpu_configs.py tests/test_gpu_edge.py::test_duplicate_heavy_user_with_more_rows_than_items_trains tests/test_gpu_api.py::test_resumed_fit_with_a_fixed_seed_does_not_replay_order_and_draws -x -q -s -m gpu ) > gpurun_out/r02a_tests.log 2>&1
tail -30 gpurun_out/r02a_tests.log
export RFM_PROFILE_PASSES=stats
bash tools/profile_bench.sh r02a_c3 --config C3 --steps 10 --warmup 5 > gpurun_out/r02a_c3.log 2>&1
bash tools/profile_bench.sh r02a_c4 --config C4 --steps 5 --warmup 2 > gpurun_out/r02a_c4.log 2>&1
bash tools/profile_bench.sh r02a_c5 --config C5 --steps 5 --warmup 3 > gpurun_out/r02a_c5.log 2>&1
python bench.py --steps 20 --warmup 3 > gpurun_out/r02a_c2_bench.json 2> gpurun_out/r02a_c2_bench.err
tail -3 gpurun_out/r02a_c3.log gpurun_out/r02a_c4.log gpurun_out/r02a_c5.log
cut -c1-400 gpurun_out/r02a_c2_bench.json
