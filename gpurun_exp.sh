R=$GRAFT_REPO_ROOT; cd /tmp; export TMPDIR=/tmp
$R/tools/pmc_calibration/calib
for c in FETCH_SIZE WRITE_SIZE "TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum TCC_EA0_ATOMIC_sum"; do
  n=$(echo $c | tr ' ' '_' | cut -c1-20)
  rocprofv3 --pmc $c --kernel-trace --output-format csv -d $R/gpurun_out/calib_$n -o pmc -- $R/tools/pmc_calibration/calib > /dev/null 2>&1 || echo fail $c
done
cd $R; python - <<'PY'
import csv,glob
for d in sorted(glob.glob('gpurun_out/calib_*/')):
    for r in csv.DictReader(open(d+'pmc_counter_collection.csv')):
        if 'gather64' in r['Kernel_Name'] or 'atomic64' in r['Kernel_Name']:
            print(r['Kernel_Name'][:12], r['Counter_Name'], r['Counter_Value'])
PY
