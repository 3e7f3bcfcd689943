#!/bin/bash
cd ${GRAFT_REPO_ROOT:-.}
O=gpurun_out/r05j; mkdir -p $O
( timeout 1500 python -m pytest tests -x -q -m gpu --deselect tests/test_gpu_quality.py ) > $O/gpu_tests.log 2>&1; tail -5 $O/gpu_tests.log
( timeout 300 python bench.py --also "" --no-cpu-baseline ) > $O/bench_c2.json 2> $O/bench.err; python - <<'PY'
import json
d=json.load(open("gpurun_out/r05j/bench_c2.json"))
print(d["value"], d["ms_per_step"], d["roofline"]["frac"], d["roofline"]["kernel_ms_median"], d["roofline"]["kernel"])
PY
