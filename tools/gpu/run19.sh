#!/bin/bash
b() { python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-secondary 2>/dev/null | grep '^{"metric"' | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); h=d['roofline_hbm']; print('$1', d['value'], d['ms_per_step'], 'photometric: avg_launch_us', h['avg_launch_us'], 'ms/step', h['ms_per_step'], 'frac', h['frac'])"; }
for i in 1 2; do
  b "waves2"
  CODEPS_HIP_LIB=$PWD/codeps_amd/variants/libcodeps_hip_photo3.so b "waves3"
  CODEPS_HIP_LIB=$PWD/codeps_amd/variants/libcodeps_hip_photo4.so b "waves4"
done
