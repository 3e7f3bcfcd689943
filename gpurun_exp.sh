timeout 900 python -m pytest tests/test_gpu_api.py -m gpu -q --timeout 600 2>&1 | grep -E "^E  |passed|failed|^FAILED" | head -10
python tools/infer_timing.py 2>&1 | grep predict
