#!/bin/bash
python -m pytest tests/test_round5_epilogue_gpu.py -x -q -m gpu -k "weight_gradient_of_1x1" 2>&1 | tail -3
CODEPS_WGRAD_1X1S=1 python tools/gpu/bench_wgrad1x1.py
