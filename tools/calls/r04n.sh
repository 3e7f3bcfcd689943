#!/bin/bash
cd ${GRAFT_REPO_ROOT:-.}
O=gpurun_out/r04n; mkdir -p $O
( timeout 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_configs.py -q -m gpu -x -k "full_size or config3" -s ) > $O/tight_tests.log 2>&1; grep -E "passed|failed|full-size config 2|config 3 vs" $O/tight_tests.log | cut -c1-400
( timeout 300 python bench.py --steps 20 --warmup 3 ) > $O/bench_c2.json 2> $O/bench_c2.err; tail -c 3000 $O/bench_c2.json
( timeout 300 python bench.py --config C3 --steps 10 --warmup 3 --no-cpu-baseline ) > $O/bench_c3.json 2> $O/bench_c3.err; tail -c 1500 $O/bench_c3.json
( timeout 300 python bench.py --config C4 --steps 10 --warmup 2 --no-cpu-baseline ) > $O/bench_c4.json 2> $O/bench_c4.err; tail -c 1500 $O/bench_c4.json
( timeout 600 python bench.py --config C5 --steps 4 --warmup 1 --no-cpu-baseline ) > $O/bench_c5.json 2> $O/bench_c5.err; tail -c 1500 $O/bench_c5.json
bash tools/profile_bench.sh r04_c2 --config C2 > $O/profile_c2.log 2>&1; tail -5 $O/profile_c2.log
RFM_PROFILE_PASSES="stats" bash tools/profile_bench.sh r04_c3 --config C3 --steps 10 > $O/profile_c3.log 2>&1; tail -3 $O/profile_c3.log
RFM_PROFILE_PASSES="stats FETCH_SIZE WRITE_SIZE TCC_EA0_ATOMIC_sum_TCC_EA0_RDREQ_sum_TCC_EA0_WRREQ_sum" bash tools/profile_bench.sh r04_c4 --config C4 --steps 10 > $O/profile_c4.log 2>&1; tail -3 $O/profile_c4.log
