#!/bin/bash
# round 4: the bench lines and rocprof summaries of the final tree
cd ${GRAFT_REPO_ROOT:-.}
O=gpurun_out/r04final; mkdir -p $O
( timeout 300 python bench.py --steps 20 --warmup 3 ) > $O/bench_c2.json 2> $O/bench_c2.err; tail -c 600 $O/bench_c2.json
( timeout 300 python bench.py --config C3 --steps 10 --warmup 3 --no-cpu-baseline ) > $O/bench_c3.json 2> $O/bench_c3.err
( timeout 300 python bench.py --config C4 --steps 10 --warmup 2 --no-cpu-baseline ) > $O/bench_c4.json 2> $O/bench_c4.err
( timeout 600 python bench.py --config C5 --steps 4 --warmup 1 --no-cpu-baseline ) > $O/bench_c5.json 2> $O/bench_c5.err
for c in c2 c3 c4 c5; do python - $O/bench_$c.json <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().split("\n")[-1])
r=d["roofline"]
print(sys.argv[1][-13:], "value %.0f M/s  kernel ms %.3f (min %.3f median %.3f)  frac %.4f  mhz %s  draws %.2f" % (d["value"]/1e6, r["kernel_ms_per_launch"], r["kernel_ms_min"], r["kernel_ms_median"], r["frac"], r["shader_mhz"], d["config"]["mean_draws_per_update"]))
PY
done
bash tools/profile_bench.sh r04f_c2 --config C2 > $O/profile_c2.log 2>&1; tail -3 $O/profile_c2.log
RFM_PROFILE_PASSES="stats FETCH_SIZE WRITE_SIZE TCC_EA0_ATOMIC_sum_TCC_EA0_RDREQ_sum_TCC_EA0_WRREQ_sum SQ_INSTS_VALU_SQ_INSTS_SALU_SQ_INSTS_LDS_SQ_WAVES" bash tools/profile_bench.sh r04f_c3 --config C3 --steps 10 > $O/profile_c3.log 2>&1; tail -3 $O/profile_c3.log
RFM_PROFILE_PASSES="stats FETCH_SIZE WRITE_SIZE TCC_EA0_ATOMIC_sum_TCC_EA0_RDREQ_sum_TCC_EA0_WRREQ_sum" bash tools/profile_bench.sh r04f_c4 --config C4 --steps 10 > $O/profile_c4.log 2>&1; tail -3 $O/profile_c4.log
RFM_PROFILE_PASSES="stats" bash tools/profile_bench.sh r04f_c5 --config C5 --steps 4 --warmup 1 > $O/profile_c5.log 2>&1; tail -3 $O/profile_c5.log
