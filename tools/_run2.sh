#!/bin/bash
# round-2 evidence run: full GPU test suite, profiles of every config, run-to-run variation, boundary timings, config-1 quality
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export RFM_DATA_CACHE=/tmp/rfmc
( time timeout 1500 python -m pytest tests -q -m gpu ) > gpurun_out/r02_tests.log 2>&1
grep -E "passed|failed|^FAILED|^E  +Assert" gpurun_out/r02_tests.log | cut -c1-200 | tail -15
bash tools/profile_bench.sh r02_c2 --steps 20 --warmup 3 > gpurun_out/r02_c2.log 2>&1
RFM_PROFILE_PASSES="stats FETCH_SIZE WRITE_SIZE TCC_EA0_ATOMIC_sum_TCC_EA0_RDREQ_sum_TCC_EA0_WRREQ_sum SQ_WAVE_CYCLES_SQ_WAIT_ANY_SQ_LDS_IDX_ACTIVE_SQ_INSTS_LDS_ATOM" bash tools/profile_bench.sh r02_c3 --config C3 --steps 10 --warmup 5 > gpurun_out/r02_c3.log 2>&1
RFM_PROFILE_PASSES="stats FETCH_SIZE WRITE_SIZE TCC_EA0_ATOMIC_sum_TCC_EA0_RDREQ_sum_TCC_EA0_WRREQ_sum SQ_INSTS_VALU_SQ_INSTS_SALU_SQ_INSTS_LDS_SQ_WAVES SQ_WAVE_CYCLES_SQ_WAIT_ANY_SQ_LDS_IDX_ACTIVE_SQ_INSTS_LDS_ATOM" bash tools/profile_bench.sh r02_c4 --config C4 --steps 5 --warmup 2 > gpurun_out/r02_c4.log 2>&1
RFM_PROFILE_PASSES="stats FETCH_SIZE WRITE_SIZE TCC_EA0_ATOMIC_sum_TCC_EA0_RDREQ_sum_TCC_EA0_WRREQ_sum" bash tools/profile_bench.sh r02_c5 --config C5 --steps 5 --warmup 3 > gpurun_out/r02_c5.log 2>&1
for k in 1 2 3 4 5 6 7 8; do python bench.py --steps 20 --warmup 3 --no-cpu-baseline | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('run', d['value']/1e9, d['roofline']['kernel_ms_per_launch'])"; done > gpurun_out/r02_c2_repeat.log 2>&1
python bench.py > gpurun_out/r02_c2_bench.json 2> /dev/null
python tools/host_path_timing.py > gpurun_out/r02_host_path.log 2>&1
python tools/movielens_quality.py 3 > gpurun_out/r02_movielens_quality.log 2>&1
python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/r02_smoke.log 2>&1; tail -n 2 gpurun_out/r02_smoke.log
cat gpurun_out/r02_c2_repeat.log; tail -n 3 gpurun_out/r02_host_path.log; tail -n 8 gpurun_out/r02_movielens_quality.log
