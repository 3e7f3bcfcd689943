#!/bin/bash
# the whole GPU suite + smoke, as the driver runs them at round end
cd ${GRAFT_REPO_ROOT:-.}
O=gpurun_out/r05suite; mkdir -p $O
( time timeout 3000 python -m pytest tests -x -q -m gpu ) > $O/gpu_suite.log 2>&1; tail -6 $O/gpu_suite.log
( timeout 600 python -c "import __graft_entry__ as g; g.smoke()" ) > $O/smoke.log 2>&1; tail -4 $O/smoke.log
