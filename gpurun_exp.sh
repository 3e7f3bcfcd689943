timeout 900 python -m pytest tests -m gpu -q --timeout 600 2>&1 | grep -E "^E  |passed|failed|^FAILED" | head -20
python tools/parity_sweep.py --users 1000000 --items 200000 --rows 50000000 --factors 64 --epochs 2 --no-oracle 2>&1 | grep -v amdgpu.ids | cut -c1-200
