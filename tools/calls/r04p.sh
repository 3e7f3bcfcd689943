#!/bin/bash
cd ${GRAFT_REPO_ROOT:-.}
O=gpurun_out/r04p; mkdir -p $O
( timeout 200 python tools/ab_kernel.py --config C2 --variants "base;flags=512;hot_publications=24,sweep_every=2;flags=512,hot_publications=24,sweep_every=2;base;flags=512" --epochs 6 --rounds 3 ) > $O/ab_c2.log 2>&1; tail -6 $O/ab_c2.log
( timeout 200 python tools/ab_kernel.py --config C2 --zipf 0 --variants "base;flags=512" --epochs 6 --rounds 3 ) > $O/ab_c2_uniform.log 2>&1; tail -2 $O/ab_c2_uniform.log
