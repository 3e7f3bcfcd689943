#!/bin/bash
# tools/pmc_once.sh "<counter names>" [bench.py arguments]: one rocprofv3 --pmc pass over bench.py (3 steps), mean per SGD launch
set -u
C=$1; shift
R=${GRAFT_REPO_ROOT:-$(pwd)}
D=$(mktemp -d /tmp/pmc.XXXX)
cd /tmp && export TMPDIR=/tmp && cd $R
timeout 300 rocprofv3 --kernel-trace --pmc $C -d $D -o out --output-format csv -- python bench.py --no-cpu-baseline --steps 3 --warmup 1 "$@" > $D/log 2>&1 || tail -5 $D/log
python - $D <<'PY'
import sys, csv, glob, collections
acc = collections.defaultdict(lambda: [0, 0.0])
for f in glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "sgd_" in r["Kernel_Name"]:
            a = acc[r["Counter_Name"]]; a[0] += 1; a[1] += float(r["Counter_Value"])
print({k: round(v / max(n, 1)) for k, (n, v) in sorted(acc.items())})
PY
