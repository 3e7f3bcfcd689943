R=$GRAFT_REPO_ROOT; cd /tmp; export TMPDIR=/tmp
for c in "SQ_WAVES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS" "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_FLAT SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_INSTS_SMEM"; do
  n=$(echo $c | tr ' ' '_' | cut -c1-30)
  rocprofv3 --pmc $c --kernel-trace --output-format csv -d $R/gpurun_out/prof_f_pmc_$n -o pmc -- python $R/bench.py --config C4S --steps 2 --warmup 1 --no-cpu-baseline > /dev/null 2>&1 || echo "pmc $c failed"
done
cd $R; python - <<'PY'
import csv,glob,collections
for d in sorted(glob.glob('gpurun_out/prof_f_pmc_*/')):
    rows=list(csv.DictReader(open(d+'pmc_counter_collection.csv')))
    agg=collections.defaultdict(list)
    for r in rows:
        if 'sgd_segments' in r['Kernel_Name']: agg[r['Counter_Name']].append(float(r['Counter_Value']))
    for k,v in agg.items(): print(k, '%.4g'%(sum(v)/len(v)))
PY
