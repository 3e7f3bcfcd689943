#!/bin/bash
cd ${GRAFT_REPO_ROOT:-.}
O=gpurun_out/r04full; mkdir -p $O
( timeout 2400 python -m pytest tests -q -m gpu -x ) > $O/gputests.log 2>&1; tail -5 $O/gputests.log
( timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" ) > $O/smoke.log 2>&1; tail -2 $O/smoke.log
