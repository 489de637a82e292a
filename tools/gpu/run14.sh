#!/bin/bash
mkdir -p gpurun_out/r5
python -m pytest tests/test_headline_parity_gpu.py -q -s 2>&1 | grep -E "parity|passed|failed|Error|assert" | head -20
python -m pytest tests/test_distributed_gpu.py tests/test_round4_ops_gpu.py -q 2>&1 | tail -3
