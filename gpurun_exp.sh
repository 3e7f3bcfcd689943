timeout 900 python -m pytest tests -m gpu -q --timeout 600 2>&1 | grep -E "^E  .*(Error|norm|corr|Mismatch|Max|assert)|passed|failed|^FAILED" | head -30
python tools/quality_parity.py --seeds 3 2>&1 | grep -E "^==|^hit|^nvu|^nvi|^nwi|^t "
