#!/bin/bash
# One GPU-box call = a chain of steps separated by "--" (tools/calls/README.md).  Runs from the repository root of the box.
cd ${GRAFT_REPO_ROOT:-.}
O=gpurun_out; mkdir -p $O
export TMPDIR=/tmp
run_step() {
    local step=$1; shift
    case $step in
        suite)
            ( time timeout 3000 python -m pytest tests -x -q -s -m gpu ) > $O/gpu_suite.log 2>&1; tail -8 $O/gpu_suite.log
            ( timeout 600 python -c "import __graft_entry__ as g; g.smoke()" ) > $O/smoke.log 2>&1; tail -4 $O/smoke.log ;;
        tests)
            local log=$O/tests_$(date +%H%M%S).log
            ( time timeout 3000 python -m pytest -q -m gpu "$@" ) > $log 2>&1; tail -25 $log ;;
        bench)
            local tag=$1; shift
            ( timeout 1200 python bench.py "$@" ) > $O/${tag}_bench.json 2> $O/${tag}_bench.err; tail -c 2500 $O/${tag}_bench.json; tail -3 $O/${tag}_bench.err ;;
        stats)
            local tag=$1; shift
            RFM_PROFILE_PASSES="stats" bash tools/profile_bench.sh $tag "$@" > $O/${tag}_profile.log 2>&1; tail -6 $O/${tag}_profile.log | cut -c1-400 ;;
        pmc)        # the kernel statistics AND the separate --pmc passes of tools/profile_bench.sh (RFM_PMC_PASSES="FETCH_SIZE WRITE_SIZE" = a subset)
            local tag=$1; shift
            RFM_PROFILE_PASSES="${RFM_PMC_PASSES:-all}" bash tools/profile_bench.sh $tag "$@" > $O/${tag}_pmc.log 2>&1; tail -3 $O/${tag}_pmc.log | cut -c1-600 ;;
        prof)       # rocprofv3 --kernel-trace --stats of any python tool: prof <tag> <script> [args] -> gpurun_out/<tag>_kernel_stats.csv
            local tag=$1; shift
            mkdir -p $O/$tag
            ( cd /tmp && rocprofv3 --kernel-trace --stats -d $OLDPWD/$O/$tag -o out --output-format csv -- python $OLDPWD/"$@" ) > $O/${tag}_prof.log 2>&1
            cp $(find $O/$tag -name "*kernel_stats.csv" | head -1) $O/${tag}_kernel_stats.csv 2>/dev/null; head -12 $O/${tag}_kernel_stats.csv | cut -c1-200; tail -4 $O/${tag}_prof.log | cut -c1-300 ;;
        py)
            local script=$1; shift
            local log=$O/$(basename $script .py)_$(date +%H%M%S).log
            ( time timeout ${RFM_STEP_TIMEOUT:-2400} python $script "$@" ) > $log 2>&1; tail -${RFM_TAIL:-40} $log ;;
        *) echo "unknown step $step"; return 2 ;;
    esac
}
args=()
for a in "$@"; do
    if [ "$a" = "--" ]; then
        [ ${#args[@]} -gt 0 ] && { echo "=== ${args[*]}"; run_step "${args[@]}"; }
        args=()
    else
        args+=("$a")
    fi
done
[ ${#args[@]} -gt 0 ] && { echo "=== ${args[*]}"; run_step "${args[@]}"; }
