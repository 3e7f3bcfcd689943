#!/bin/bash
# round 4, GPU call A: row-gather microbenchmark, parity of the dynamic segment order / pipelined sweeps, A/B timings
cd ${GRAFT_REPO_ROOT:-.}
O=gpurun_out/r04a; mkdir -p $O
( timeout 120 tools/microbench/row_gather ) > $O/row_gather.log 2>&1
( timeout 600 python -m pytest tests/test_gpu_parity.py -x -q -m gpu ) > $O/parity.log 2>&1; tail -3 $O/parity.log
( timeout 300 python tools/ab_kernel.py --config C2 --variants "base;flags=128;flags=256;flags=384" --epochs 6 --rounds 3 ) > $O/ab_c2.log 2>&1; tail -4 $O/ab_c2.log
( timeout 300 python tools/ab_kernel.py --config C4 --variants "base;flags=128;flags=256;flags=384" --epochs 4 --rounds 2 ) > $O/ab_c4.log 2>&1; tail -4 $O/ab_c4.log
( timeout 300 python tools/ab_kernel.py --config C3 --variants "base;flags=128;damping=256;damping=512;damping=1024" --warmup 0 --epochs 4 --rounds 1 --print-ll ) > $O/ab_c3.log 2>&1; tail -10 $O/ab_c3.log
( timeout 300 python tools/ab_kernel.py --config C2 --variants "base;damping=256;damping=512;damping=1024" --warmup 0 --epochs 4 --rounds 1 --print-ll ) > $O/ab_c2_damp.log 2>&1; tail -8 $O/ab_c2_damp.log
cat $O/row_gather.log
