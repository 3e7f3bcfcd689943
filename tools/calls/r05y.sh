#!/bin/bash
cd ${GRAFT_REPO_ROOT:-.}
O=gpurun_out/r05y; mkdir -p $O
for mode in auto late blocking; do
( RFM_BENCH_SHARE_GPU=1 timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29517 bench.py --gpus 2 --steps 3 --warmup 2 --no-strong --exchange $mode ) > $O/bench_n2_$mode.json 2> $O/bench_n2_$mode.err
python - $O/bench_n2_$mode.json <<'PY'
import json, sys
try:
    d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    print(d["n_gpus"], d["value"], d["ms_per_step"], d["config"]["exchange"])
except Exception as e:
    print("FAILED", e); print(open(sys.argv[1].replace(".json", ".err")).read()[-1500:])
PY
done
