#!/bin/bash
cd ${GRAFT_REPO_ROOT:-.}
O=gpurun_out/r05f; mkdir -p $O
( time timeout 900 python bench.py ) > $O/bench_default.json 2> $O/bench_default.err; tail -c 3000 $O/bench_default.json; tail -5 $O/bench_default.err
