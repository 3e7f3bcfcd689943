timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -q --timeout 600 -k two_user_shards 2>&1 | grep -E "^E  |passed|failed" | head -6
