#!/bin/bash
# Regenerates the profile artifacts of one round on the GPU box (run through gpurun from the repository root):
#   tools/profile_bench.sh <tag> [bench.py arguments]      (RFM_PROFILE_PASSES="stats" = kernel statistics only; "stats FETCH_SIZE ..." = a subset)
# writes gpurun_out/<tag>_kernel_stats.csv, <tag>_bench_under_rocprof.json and <tag>_pmc.json; copy them into profiles/.
# The counter passes are separate runs with --kernel-trace only (gpurun refuses --pmc together with other trace domains);
# FETCH_SIZE and WRITE_SIZE each need a pass of their own (together: "exceeds the capabilities of the hardware to collect").
set -u
TAG=$1; shift
PASSES=${RFM_PROFILE_PASSES:-all}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out
mkdir -p $OUT/$TAG
cd /tmp && export TMPDIR=/tmp
cd $R
if [ "$PASSES" = all ] || echo " $PASSES " | grep -q " stats "; then
rocprofv3 --kernel-trace --stats -d $OUT/$TAG/stats -o out --output-format csv -- python bench.py --no-cpu-baseline "$@" > $OUT/${TAG}_bench_under_rocprof.json 2> $OUT/$TAG/stats.log
cp $(find $OUT/$TAG/stats -name "*kernel_stats.csv" | head -1) $OUT/${TAG}_kernel_stats.csv
fi
for C in "FETCH_SIZE" "WRITE_SIZE" "TCC_EA0_ATOMIC_sum TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum" "TCC_HIT_sum TCC_MISS_sum TCC_ATOMIC_sum" \
         "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVES" "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS_ATOMIC"; do
    n=$(echo $C | tr ' ' '_' | cut -c1-48)
    if [ "$PASSES" != all ] && ! echo " $PASSES " | grep -q " $n "; then continue; fi     # e.g. RFM_PROFILE_PASSES="FETCH_SIZE WRITE_SIZE"
    timeout 300 rocprofv3 --kernel-trace --pmc $C -d $OUT/$TAG/$n -o out --output-format csv -- python bench.py --no-cpu-baseline --steps 3 --warmup 1 "$@" > $OUT/$TAG/$n.log 2>&1
done
python - $OUT/$TAG $OUT/${TAG}_pmc.json <<'PY'
import sys, csv, glob, json, collections
acc = collections.defaultdict(lambda: [0, 0.0])
for f in glob.glob(sys.argv[1] + "/*/**/*counter_collection.csv", recursive=True) + glob.glob(sys.argv[1] + "/*/*counter_collection.csv"):
    seen = set()
    for r in csv.DictReader(open(f)):
        # per SGD launch: the row-loop kernel's counters + (models with features) those of the tables kernel that runs beside it
        if "sgd_" in r["Kernel_Name"] or "feat_tables_kernel" in r["Kernel_Name"]:
            a = acc[r["Counter_Name"]]; a[1] += float(r["Counter_Value"])
            if "sgd_" in r["Kernel_Name"]: a[0] += 1
out = {k: {"launches": n, "mean_per_launch": v / max(n, 1)} for k, (n, v) in sorted(acc.items())}
json.dump(out, open(sys.argv[2], "w"), indent=1)
print(json.dumps({k: v["mean_per_launch"] for k, v in out.items()}))
PY
tail -1 $OUT/${TAG}_bench_under_rocprof.json | cut -c1-300
head -4 $OUT/${TAG}_kernel_stats.csv
