#!/bin/bash
export TMPDIR=/tmp
mkdir -p gpurun_out/r5
rocprofv3 --kernel-trace --output-format csv -d gpurun_out/prof_t -o r -- python3 bench.py --steps 8 --warmup 3 --no-cpu-baseline --no-roofline --no-secondary > /dev/null 2>&1
python3 tools/trace_steps.py gpurun_out/prof_t 2 --hist > gpurun_out/r5/trace_now.json 2>/dev/null
rm -rf gpurun_out/prof_t
