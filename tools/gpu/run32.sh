#!/bin/bash
mkdir -p gpurun_out/r5
python -m pytest tests/test_losses_gpu.py tests/test_losses_fuzz_gpu.py tests/test_configs_gpu.py tests/test_graph_gpu.py tests/test_network_gpu.py -x -q -m gpu > gpurun_out/r5/t32.log 2>&1
echo "tests rc=$?"; tail -5 gpurun_out/r5/t32.log
python bench.py --steps 30 --warmup 8 --no-secondary --no-cpu-baseline --no-roofline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print('adapt', d['value'], d['ms_per_step'])"
