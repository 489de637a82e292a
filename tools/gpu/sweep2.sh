#!/bin/bash
# env-knob sweep of the conv engine's launch plans on the headline step under the fp16-pair arithmetic (same box, interleaved with the default)
run() { env "$@" python bench.py --steps 30 --warmup 8 --no-secondary --no-cpu-baseline --no-roofline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print('$*', d['ms_per_step'], d['step_ms']['median'])"; }
run X=0
run CODEPS_WS_MIN=16
run CODEPS_WS_MIN=24
run CODEPS_WS_MIN=64
run X=0
run CODEPS_PENS=0.35
run CODEPS_PENS=0.7
run CODEPS_PEN1=0.9
run CODEPS_PEN1=1.3
run X=0
run CODEPS_SHORTK=15
run CODEPS_SHORTK=47
run CODEPS_HALO3=1
run CODEPS_HALO3=4
run X=0
run CODEPS_SPLITK_BLOCKS=128
run CODEPS_SPLITK_TARGET=384
run CODEPS_HALO_RING=0
run CODEPS_RING8=0
run X=0
