#!/bin/bash
# round 5, after the WARP kernels' 8-deep register lists: bench + stats of configs 3 and 5 again, then the whole suite
cd ${GRAFT_REPO_ROOT:-.}
O=gpurun_out
( timeout 900 python bench.py ) > $O/r05_bench.json 2> $O/r05_bench.err; tail -c 300 $O/r05_bench.json
RFM_PROFILE_PASSES="stats FETCH_SIZE WRITE_SIZE" bash tools/profile_bench.sh r05_c3 --config C3 > $O/r05_c3_profile.log 2>&1; tail -3 $O/r05_c3_profile.log | cut -c1-300
RFM_PROFILE_PASSES="stats FETCH_SIZE WRITE_SIZE" bash tools/profile_bench.sh r05_c5 --config C5 --steps 5 --warmup 2 > $O/r05_c5_profile.log 2>&1; tail -3 $O/r05_c5_profile.log | cut -c1-300
bash tools/calls/r05suite.sh
