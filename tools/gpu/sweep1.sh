#!/bin/bash
# env-knob sweep of the conv engine's launch plans on the headline step (same box, interleaved with the default)
run() { env "$@" python bench.py --steps 30 --warmup 8 --no-secondary --no-cpu-baseline --no-roofline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print('$*', d['ms_per_step'], d['step_ms']['median'])"; }
run X=0
run CODEPS_SPLITK_BLOCKS=128
run CODEPS_SPLITK_BLOCKS=256
run X=0
run CODEPS_SPLITK_TARGET=384
run CODEPS_SPLITK_TARGET=768
run X=0
run CODEPS_PENS=0.4
run CODEPS_PENS=0.65
run CODEPS_WS_MIN=24
run CODEPS_WS_MIN=48
run X=0
run CODEPS_SHORTK=23
run CODEPS_SHORTK=39
run CODEPS_PEN1=1.0
run CODEPS_PEN1=1.25
run X=0
