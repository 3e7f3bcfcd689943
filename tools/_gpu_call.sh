export OMP_NUM_THREADS=8 OPENBLAS_NUM_THREADS=8 MKL_NUM_THREADS=8
for i in 1 2; do timeout 300 python -m pytest tests -q -m gpu -x -s -k "feature and not config4" 2>&1 | grep -v "^$" | tail -9; done
FQ_RUNS=6 timeout 200 python tools/feature_quality.py 2>&1 | tail -6
