#!/bin/bash
# round 5, after the WARP kernel's deferred atomics: the bench line and config 3's profile of the final tree
cd ${GRAFT_REPO_ROOT:-.}
O=gpurun_out
( timeout 900 python bench.py ) > $O/r05_bench.json 2> $O/r05_bench.err; tail -c 1500 $O/r05_bench.json
RFM_PROFILE_PASSES="stats FETCH_SIZE WRITE_SIZE" bash tools/profile_bench.sh r05_c3 --config C3 > $O/r05_c3_profile.log 2>&1; tail -3 $O/r05_c3_profile.log | cut -c1-300
