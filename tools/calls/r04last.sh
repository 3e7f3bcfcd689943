#!/bin/bash
cd ${GRAFT_REPO_ROOT:-.}
O=gpurun_out/r04last; mkdir -p $O
timeout 100 python -m pytest tests/test_gpu_parity.py -q -x -m gpu -k "golden or statistical or feature" > $O/pytest.log 2>&1; tail -2 $O/pytest.log
timeout 60 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -1
