#!/bin/bash
# end-of-round evidence after the row-loop trims and the longer stripe window: config-2 parity at full size, profile, repeats, boundary
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export RFM_DATA_CACHE=/tmp/rfmc
timeout 200 python -m pytest tests/test_gpu_parity.py -q -s -k "full_size_config2 or conserves" 2>&1 | grep -E "full-size|passed|failed" | cut -c1-250
RFM_PROFILE_PASSES="stats FETCH_SIZE WRITE_SIZE" timeout 200 bash tools/profile_bench.sh r02_c2 --steps 20 --warmup 3 > gpurun_out/r02_c2.log 2>&1; tail -n 3 gpurun_out/r02_c2.log | cut -c1-300
for k in 1 2 3 4; do timeout 60 python bench.py --steps 20 --warmup 3 --no-cpu-baseline | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('run', d['value']/1e9, d['roofline']['kernel_ms_per_launch'], d['config']['final_mean_ll_per_update'])"; done > gpurun_out/r02_c2_repeat.log 2>&1
cat gpurun_out/r02_c2_repeat.log
timeout 120 python bench.py > gpurun_out/r02_c2_bench.json 2> /dev/null; cut -c1-400 gpurun_out/r02_c2_bench.json
timeout 100 python tools/host_path_timing.py 2>&1 | grep -v amdgpu.ids > gpurun_out/r02_host_path.log; tail -n 6 gpurun_out/r02_host_path.log
for c in C3 C4; do timeout 120 python bench.py --config $c --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$c', d['value']/1e6, d['roofline']['kernel_ms_per_launch'], d['roofline']['frac'])"; done
