python -m pytest tests/test_round5_epilogue_gpu.py -q -s -k "batchnorm" 2>&1 | grep -E "BN case|passed|failed|AssertionError" | head -20
echo ---- single cases
for k in case1 case4; do python -m pytest tests/test_round5_epilogue_gpu.py -q -s -k "batchnorm and $k" 2>&1 | grep -E "BN case|passed|failed|AssertionError" | head -5; done
