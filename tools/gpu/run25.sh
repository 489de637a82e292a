#!/bin/bash
mkdir -p gpurun_out/r5
python -m pytest tests/test_round5_epilogue_gpu.py tests/test_modules_gpu.py tests/test_configs_gpu.py tests/test_graph_gpu.py -x -q -m gpu > gpurun_out/r5/t25.log 2>&1
echo "tests rc=$?"; tail -3 gpurun_out/r5/t25.log
for i in 1 2; do
  python bench.py --mode train --steps 12 --warmup 4 --no-secondary --no-cpu-baseline --no-roofline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print('pretrain', d['value'], d['ms_per_step'])"
done
python bench.py --steps 30 --warmup 8 --no-secondary --no-cpu-baseline --no-roofline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print('adapt', d['value'], d['ms_per_step'])"
