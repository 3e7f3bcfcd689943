#!/bin/bash
cd ${GRAFT_REPO_ROOT:-.}
O=gpurun_out/r05l; mkdir -p $O
( timeout 600 python -m pytest tests/test_gpu_distributed.py -x -q -m gpu ) > $O/dist.log 2>&1; tail -4 $O/dist.log
( RFM_BENCH_SHARE_GPU=1 timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 2 --steps 4 --warmup 2 --no-strong ) > $O/bench_n2_shared.json 2> $O/bench_n2.err; tail -c 1800 $O/bench_n2_shared.json; tail -3 $O/bench_n2.err
( timeout 1200 python -m pytest tests/test_gpu_quality.py -x -q -m gpu -s -k "eight_engine" ) > $O/shards.log 2>&1; grep -E "eight engine|passed|failed|Error" $O/shards.log | cut -c1-400
