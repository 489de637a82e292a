#!/bin/bash
mkdir -p gpurun_out/r5
python -m pytest tests -x -q -m gpu > gpurun_out/r5/t34.log 2>&1
echo "all tests rc=$?"; tail -4 gpurun_out/r5/t34.log
for s in 1 0 1 0; do
CODEPS_DEFER_COLSUM=$s python bench.py --steps 30 --warmup 8 --no-secondary --no-cpu-baseline --no-roofline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print('adapt defer=$s', d['value'], d['ms_per_step'])"
done
