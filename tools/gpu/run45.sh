#!/bin/bash
export TMPDIR=/tmp
mkdir -p gpurun_out/r5
rocprofv3 --kernel-trace --output-format csv -d gpurun_out/prof_t -o r -- python3 bench.py --steps 8 --warmup 3 --no-cpu-baseline --no-roofline --no-secondary > /dev/null 2>&1
head -1 $(find gpurun_out/prof_t -name "*kernel_trace.csv" | head -1)
python3 tools/critical_path.py gpurun_out/prof_t > gpurun_out/r5/critical_path.json
rm -rf gpurun_out/prof_t
