#!/bin/bash
# round 4, final tree: the whole -m gpu suite, smoke, then the bench lines and rocprof summaries
cd ${GRAFT_REPO_ROOT:-.}
bash tools/calls/r04suite.sh
bash tools/calls/r04final.sh
