#!/bin/bash
cd ${GRAFT_REPO_ROOT:-.}
O=gpurun_out/r05h; mkdir -p $O
( timeout 400 python tools/ab_kernel.py --config C2 --variants "base;flags=1024;flags=512;flags=1536;base;flags=1024;hot_publications=24;flags=1024,hot_publications=24;hot_publications=32" --epochs 5 --rounds 3 ) > $O/ab_c2.log 2>&1; tail -10 $O/ab_c2.log
( timeout 600 python -m pytest tests/test_gpu_parity.py -x -q -m gpu ) > $O/parity.log 2>&1; tail -3 $O/parity.log
