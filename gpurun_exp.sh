P='import sys,json; d=json.loads(sys.stdin.read()); print("%.1f M/s  %.3f ms  ll %.4f frac %.3f draws %.2f" % (d["value"]/1e6, d["roofline"]["kernel_ms_per_launch"], d["config"]["final_mean_ll_per_update"], d["roofline"]["frac"], d["config"]["mean_draws_per_update"]))'
timeout 900 python -m pytest tests -m gpu -q --timeout 600 2>&1 | grep -E "^E  |passed|failed|^FAILED" | head -20
echo -n "C3 warp zipf: "; python bench.py --config C3 --steps 5 --warmup 2 --no-cpu-baseline 2>/dev/null | python -c "$P"
echo -n "C3 warp uniform: "; python bench.py --config C3 --steps 5 --warmup 2 --no-cpu-baseline --zipf 0 2>/dev/null | python -c "$P"
echo -n "C2 zipf: "; python bench.py --steps 10 --warmup 2 --no-cpu-baseline 2>/dev/null | python -c "$P"
