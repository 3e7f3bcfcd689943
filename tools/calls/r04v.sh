#!/bin/bash
cd ${GRAFT_REPO_ROOT:-.}
O=gpurun_out/r04v; mkdir -p $O
( timeout 900 python -m pytest tests/test_gpu_distributed.py -q -m gpu -x ) > $O/dist_tests.log 2>&1; tail -3 $O/dist_tests.log
( timeout 1500 python -m pytest tests/test_gpu_quality.py -q -m gpu -s -k "eight_engine" ) > $O/emul.log 2>&1; grep -E "eight engine shards|passed|failed|Error" $O/emul.log | cut -c1-400
( timeout 300 env RFM_BENCH_SHARE_GPU=1 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29517 bench.py --gpus 2 --steps 3 --warmup 1 --no-strong ) > $O/bench_2ranks_shared.json 2> $O/bench_2ranks.err; tail -c 1200 $O/bench_2ranks_shared.json; tail -3 $O/bench_2ranks.err
