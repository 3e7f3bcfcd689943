P='import sys,json; d=json.loads(sys.stdin.read()); print("%.1f M/s  %.3f ms frac %.3f" % (d["value"]/1e6, d["roofline"]["kernel_ms_per_launch"], d["roofline"]["frac"]))'
for f in 128 256; do echo -n "zipf F=$f: "; python bench.py --steps 5 --warmup 2 --no-cpu-baseline --factors $f 2>/dev/null | python -c "$P"; done
