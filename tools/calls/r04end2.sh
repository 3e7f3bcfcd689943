#!/bin/bash
# final tree: whole -m gpu suite + smoke, then config 4's bench line and rocprof summaries again (its quota changed after r04final)
cd ${GRAFT_REPO_ROOT:-.}
bash tools/calls/r04suite.sh
O=gpurun_out/r04final; mkdir -p $O
( timeout 300 python bench.py --config C4 --steps 10 --warmup 2 --no-cpu-baseline ) > $O/bench_c4.json 2> $O/bench_c4.err
python - $O/bench_c4.json <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().split("\n")[-1])
r=d["roofline"]
print("C4 value %.0f M/s  kernel ms %.3f (min %.3f median %.3f)  frac %.4f  mhz %s" % (d["value"]/1e6, r["kernel_ms_per_launch"], r["kernel_ms_min"], r["kernel_ms_median"], r["frac"], r["shader_mhz"]))
PY
RFM_PROFILE_PASSES="stats FETCH_SIZE WRITE_SIZE TCC_EA0_ATOMIC_sum_TCC_EA0_RDREQ_sum_TCC_EA0_WRREQ_sum" bash tools/profile_bench.sh r04f_c4 --config C4 --steps 10 > $O/profile_c4.log 2>&1; tail -3 $O/profile_c4.log
