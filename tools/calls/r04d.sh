#!/bin/bash
cd ${GRAFT_REPO_ROOT:-.}
O=gpurun_out/r04d; mkdir -p $O
( timeout 200 python tools/ab_kernel.py --config C2 --variants "flags=896;flags=640;flags=768;flags=512;flags=384;flags=896;flags=512" --warmup 0 --epochs 4 --rounds 1 --print-ll ) > $O/ab_c2_bisect.log 2>&1; tail -14 $O/ab_c2_bisect.log
