python -m pytest tests/test_headline_parity_gpu.py -q -s 2>&1 | grep -E "parity|passed|failed|Error|assert" | head -20
