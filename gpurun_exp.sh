R=$GRAFT_REPO_ROOT; cd /tmp; export TMPDIR=/tmp
P='import sys,json; d=json.loads(sys.stdin.read()); print("%.1f M/s  %.3f ms x%d launches  ll %.4f frac %.3f draws %.2f" % (d["value"]/1e6, d["roofline"]["kernel_ms_per_launch"], d["config"]["sgd_launches_per_epoch"], d["config"]["final_mean_ll_per_update"], d["roofline"]["frac"], d["config"]["mean_draws_per_update"]))'
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_b_stats -o bench -- python $R/bench.py --steps 20 --warmup 3 > $R/gpurun_out/prof_b_bench.json 2>/dev/null
for c in FETCH_SIZE WRITE_SIZE "TCC_HIT_sum TCC_MISS_sum" "TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum TCC_EA0_ATOMIC_sum" "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAIT_ANY SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_SALU"; do
  n=$(echo $c | tr ' ' '_' | cut -c1-30)
  rocprofv3 --pmc $c --kernel-trace --output-format csv -d $R/gpurun_out/prof_b_pmc_$n -o pmc -- python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline > /dev/null 2>&1 || echo "pmc $c failed"
  rocprofv3 --pmc $c --kernel-trace --output-format csv -d $R/gpurun_out/prof_b_uni_pmc_$n -o pmc -- python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --zipf 0 > /dev/null 2>&1 || echo "pmc $c failed"
done
cd $R
cat gpurun_out/prof_b_bench.json
echo -n "C2 uniform: "; python bench.py --steps 20 --warmup 3 --no-cpu-baseline --zipf 0 2>/dev/null | python -c "$P"
echo -n "C3 warp zipf: "; python bench.py --config C3 --steps 5 --warmup 2 --no-cpu-baseline 2>/dev/null | python -c "$P"
echo -n "C3 warp uniform: "; python bench.py --config C3 --steps 5 --warmup 2 --no-cpu-baseline --zipf 0 2>/dev/null | python -c "$P"
python tools/host_path_timing.py 2>&1 | grep host-buffer
