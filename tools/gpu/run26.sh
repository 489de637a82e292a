#!/bin/bash
mkdir -p gpurun_out/r5
python -m pytest tests/test_round5_epilogue_gpu.py -x -q -m gpu -k "weight_gradient_of_1x1" > gpurun_out/r5/t26.log 2>&1
echo "tests rc=$?"; tail -12 gpurun_out/r5/t26.log
for i in 1 2; do
  for s in 1 0; do
    CODEPS_WGRAD_1X1S=$s python bench.py --mode train --steps 12 --warmup 4 --no-secondary --no-cpu-baseline --no-roofline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print('pretrain w1=$s', d['value'], d['ms_per_step'])"
  done
done
for s in 1 0; do
CODEPS_WGRAD_1X1S=$s python bench.py --steps 30 --warmup 8 --no-secondary --no-cpu-baseline --no-roofline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print('adapt w1=$s', d['value'], d['ms_per_step'])"
done
