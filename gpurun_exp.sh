timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -q --timeout 600 -k conserves 2>&1 | grep -E "^E  |passed|failed|Error" | head -10
