R=$GRAFT_REPO_ROOT; cd /tmp; export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_c_stats -o bench -- python $R/bench.py --steps 20 --warmup 3 > $R/gpurun_out/prof_c_bench.json 2>/dev/null
for c in FETCH_SIZE WRITE_SIZE "TCC_HIT_sum TCC_MISS_sum" "TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum TCC_EA0_ATOMIC_sum" "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAIT_ANY SQ_WAVE_CYCLES SQ_INSTS_LDS SQ_INSTS_SALU"; do
  n=$(echo $c | tr ' ' '_' | cut -c1-30)
  rocprofv3 --pmc $c --kernel-trace --output-format csv -d $R/gpurun_out/prof_c_pmc_$n -o pmc -- python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline > /dev/null 2>&1 || echo "pmc $c failed"
done
cd $R; cat gpurun_out/prof_c_bench.json; python tools/infer_timing.py 2>&1 | grep predict
