#!/bin/bash
export TMPDIR=/tmp
mkdir -p gpurun_out/r5
python -m pytest tests/test_round5_epilogue_gpu.py -x -q -m gpu > gpurun_out/r5/t23.log 2>&1
echo "tests rc=$?"; tail -3 gpurun_out/r5/t23.log
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_pre -o r -- python3 bench.py --mode train --steps 10 --warmup 3 --no-cpu-baseline --no-roofline --no-secondary > gpurun_out/r5/pre_bench.log 2>&1
cp $(find gpurun_out/prof_pre -name "*kernel_stats.csv" | head -1) gpurun_out/r5/pretrain_kernel_stats_fin.csv
python3 tools/trace_steps.py gpurun_out/prof_pre 1 --hist > gpurun_out/r5/pretrain_step_trace_fin.json 2>/dev/null
rm -rf gpurun_out/prof_pre
