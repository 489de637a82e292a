#!/bin/bash
run() { env "$@" python bench.py $MODE --steps 20 --warmup 6 --no-secondary --no-cpu-baseline --no-roofline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print('$MODE $*', d['ms_per_step'], d['step_ms']['median'])"; }
MODE="--mode train"
run X=0
run CODEPS_DEFER_COLSUM=0
run CODEPS_FRAMES_MULTI=0
run X=0
run CODEPS_DEFER_COLSUM=0 CODEPS_FRAMES_MULTI=0
MODE=""
run X=0
run CODEPS_DEFER_COLSUM=0
run CODEPS_FRAMES_MULTI=0
run X=0
run CODEPS_DEFER_COLSUM=0 CODEPS_FRAMES_MULTI=0
