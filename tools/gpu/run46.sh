#!/bin/bash
mkdir -p gpurun_out/r5
python -m pytest tests/test_round5_epilogue_gpu.py tests/test_round4_ops_gpu.py tests/test_precision_gpu.py tests/test_configs_gpu.py tests/test_graph_gpu.py tests/test_ops_gpu.py tests/test_modules_gpu.py -x -q -m gpu > gpurun_out/r5/t46.log 2>&1
echo "tests rc=$?"; tail -3 gpurun_out/r5/t46.log
PREV=$PWD/tools/gpu/ab/libcodeps_hip_prev.so
run() { env "$@" python bench.py $MODE --steps 20 --warmup 6 --no-secondary --no-cpu-baseline --no-roofline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print('$MODE $1', d['ms_per_step'], d['step_ms']['median'])"; }
MODE=""
run NEW=1
run CODEPS_HIP_LIB=$PREV
run NEW=1
run CODEPS_HIP_LIB=$PREV
run CODEPS_REDUCE9_FINE=0
