#!/bin/bash
mkdir -p gpurun_out/r5
for i in 1 2; do
  for s in 1 0; do
    CODEPS_WGRAD_1X1S=$s python bench.py --mode train --steps 12 --warmup 4 --no-secondary --no-cpu-baseline --no-roofline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print('pretrain w1=$s', d['value'], d['ms_per_step'])"
  done
done
for s in 1 0; do
CODEPS_WGRAD_1X1S=$s python bench.py --steps 30 --warmup 8 --no-secondary --no-cpu-baseline --no-roofline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print('adapt w1=$s', d['value'], d['ms_per_step'])"
done
python -m pytest tests -x -q -m gpu > gpurun_out/r5/t28.log 2>&1
echo "all tests rc=$?"; tail -4 gpurun_out/r5/t28.log
