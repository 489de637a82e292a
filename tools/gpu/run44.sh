#!/bin/bash
mkdir -p gpurun_out/r5
python -m pytest tests -x -q -m gpu > gpurun_out/r5/t44.log 2>&1
echo "all tests rc=$?"; tail -3 gpurun_out/r5/t44.log
run() { env "$@" python bench.py $MODE --steps 20 --warmup 6 --no-secondary --no-cpu-baseline --no-roofline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print('$MODE $1', d['ms_per_step'], d['step_ms']['median'])"; }
MODE=""
run CODEPS_LOSS_STREAMS=1
run CODEPS_LOSS_STREAMS=0
run CODEPS_LOSS_STREAMS=1
run CODEPS_LOSS_STREAMS=0
