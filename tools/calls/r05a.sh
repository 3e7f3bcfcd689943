#!/bin/bash
# round 5, call a: the row-loop forms microbenchmark + this box's bench line
cd ${GRAFT_REPO_ROOT:-.}
O=gpurun_out/r05a; mkdir -p $O
( timeout 120 tools/microbench/pipe_model ) > $O/pipe_model.log 2>&1; cat $O/pipe_model.log
( timeout 300 python bench.py --steps 10 --warmup 3 ) > $O/bench_c2.json 2> $O/bench_c2.err; tail -c 1500 $O/bench_c2.json
