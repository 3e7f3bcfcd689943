#!/bin/bash
# the whole -m gpu suite + smoke on the current tree
cd ${GRAFT_REPO_ROOT:-.}
O=gpurun_out/r04suite; mkdir -p $O
timeout 900 python -m pytest tests -q -m gpu > $O/pytest.log 2>&1; tail -4 $O/pytest.log
timeout 200 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $O/smoke.log 2>&1; tail -2 $O/smoke.log
