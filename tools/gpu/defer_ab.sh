#!/bin/bash
# kernel-time A/B of CODEPS_BN_DEFER on the pre-training step (configs[1]): rocprofv3 kernel stats of both settings on one box
export TMPDIR=/tmp
for v in 1 0; do
  CODEPS_BN_DEFER=$v rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_d$v -o r -- python3 bench.py --mode train --steps 10 --warmup 3 --no-cpu-baseline --no-roofline --no-secondary > /dev/null 2>&1
  cp $(find gpurun_out/prof_d$v -name "*kernel_stats.csv" | head -1) gpurun_out/r06_configs1_kernel_stats_defer$v.csv
  rm -rf gpurun_out/prof_d$v
done
