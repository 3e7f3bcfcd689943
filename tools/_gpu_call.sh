export OMP_NUM_THREADS=8 OPENBLAS_NUM_THREADS=8 MKL_NUM_THREADS=8
mkdir -p gpurun_out/r03y
O=gpurun_out/r03y
timeout 400 python tools/ll_margins.py --runs 3 --configs C4 > $O/llm4b.log 2>&1
grep "C4:e1\|C4 run" $O/llm4b.log | cut -c1-260
timeout 300 python -m pytest tests -q -m gpu -x -s -k "feature and not config4" 2>&1 | grep -v "^$" | tail -12
