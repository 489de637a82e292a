#!/bin/bash
mkdir -p gpurun_out/r5
python -m pytest tests/test_losses_gpu.py tests/test_losses_fuzz_gpu.py tests/test_configs_gpu.py tests/test_network_gpu.py tests/test_modules_gpu.py -x -q -m gpu > gpurun_out/r5/t43.log 2>&1
echo "tests rc=$?"; tail -3 gpurun_out/r5/t43.log
PREV=$PWD/tools/gpu/ab/libcodeps_hip_prev.so
run() { env "$@" python bench.py $MODE --steps 20 --warmup 6 --no-secondary --no-cpu-baseline --no-roofline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print('$MODE $1', d['ms_per_step'], d['step_ms']['median'])"; }
MODE="--mode train"
run NEW=1
run CODEPS_HIP_LIB=$PREV
run NEW=1
run CODEPS_HIP_LIB=$PREV
