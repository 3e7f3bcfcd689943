#!/bin/bash
# round 5, the final tree: suite + smoke, then the bench line and config 2's profile once more
cd ${GRAFT_REPO_ROOT:-.}
O=gpurun_out
bash tools/calls/r05suite.sh
( timeout 900 python bench.py ) > $O/r05_bench_final.json 2> $O/r05_bench_final.err; tail -c 300 $O/r05_bench_final.json
RFM_PROFILE_PASSES="stats" bash tools/profile_bench.sh r05_c2final --also "" > $O/r05_c2final_profile.log 2>&1; tail -2 $O/r05_c2final_profile.log | cut -c1-300
