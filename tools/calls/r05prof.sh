#!/bin/bash
# round 5: the profile artifacts of the final tree (copy gpurun_out/r05_* into profiles/)
cd ${GRAFT_REPO_ROOT:-.}
O=gpurun_out
( timeout 900 python bench.py ) > $O/r05_bench.json 2> $O/r05_bench.err; tail -c 600 $O/r05_bench.json
bash tools/profile_bench.sh r05_c2 --also "" > $O/r05_c2_profile.log 2>&1; tail -4 $O/r05_c2_profile.log | cut -c1-400
RFM_PROFILE_PASSES="stats FETCH_SIZE WRITE_SIZE" bash tools/profile_bench.sh r05_c3 --config C3 > $O/r05_c3_profile.log 2>&1; tail -3 $O/r05_c3_profile.log | cut -c1-300
RFM_PROFILE_PASSES="stats FETCH_SIZE WRITE_SIZE" bash tools/profile_bench.sh r05_c4 --config C4 > $O/r05_c4_profile.log 2>&1; tail -3 $O/r05_c4_profile.log | cut -c1-300
RFM_PROFILE_PASSES="stats FETCH_SIZE WRITE_SIZE" bash tools/profile_bench.sh r05_c5 --config C5 --steps 5 --warmup 2 > $O/r05_c5_profile.log 2>&1; tail -3 $O/r05_c5_profile.log | cut -c1-300
