#!/bin/bash
# round 5: kernel statistics of the pre-training configuration (configs[1]) -- where its non-convolution third goes
export TMPDIR=/tmp
mkdir -p gpurun_out/r5
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_pre -o r -- python3 bench.py --mode train --steps 10 --warmup 3 --no-cpu-baseline --no-roofline --no-secondary > gpurun_out/r5/pre_bench.log 2>&1
cp $(find gpurun_out/prof_pre -name "*kernel_stats.csv" | head -1) gpurun_out/r5/pretrain_kernel_stats.csv
python3 tools/trace_steps.py gpurun_out/prof_pre > gpurun_out/r5/pretrain_step_trace.json 2>/dev/null
rm -rf gpurun_out/prof_pre
tail -2 gpurun_out/r5/pre_bench.log | cut -c1-300
