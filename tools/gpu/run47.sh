#!/bin/bash
PREV=$PWD/tools/gpu/ab/libcodeps_hip_prev.so
run() { env "$@" python bench.py $MODE --steps 20 --warmup 6 --no-secondary --no-cpu-baseline --no-roofline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print('$MODE $1', d['ms_per_step'], d['step_ms']['median'])"; }
MODE=""
for i in 1 2 3; do
run NEW=1
run CODEPS_HIP_LIB=$PREV
done
