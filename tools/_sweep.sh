timeout 800 python -m pytest tests -m gpu -x -q 2>&1 | tail -4
for z in 0 1.0; do
timeout 300 python bench.py --no-cpu-baseline --steps 20 --zipf $z 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('zipf', $z, d['value']/1e6, d['roofline']['kernel_ms_per_launch'], d['config']['final_mean_ll_per_update'])"
done
timeout 300 python bench.py --no-cpu-baseline --steps 5 --config C3 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('C3', d['value']/1e6, d['roofline']['kernel_ms_per_launch'], d['config']['mean_draws_per_update'])"
