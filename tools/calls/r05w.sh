#!/bin/bash
cd ${GRAFT_REPO_ROOT:-.}
AB_REPS=3 bash tools/calls/ab_builds.sh r05w_c2 --config C2 --variants "base" --epochs 5 --rounds 3 | cut -c1-150
