#!/bin/bash
# round 5: epilogue extras in the split-K finish -- tests + A/B on the pre-training configuration and the headline step
mkdir -p gpurun_out/r5
python -m pytest tests/test_round5_epilogue_gpu.py tests/test_configs_gpu.py -x -q -m gpu > gpurun_out/r5/t22.log 2>&1
echo "tests rc=$?"; tail -5 gpurun_out/r5/t22.log
for i in 1 2; do
  for s in 1 0; do
    CODEPS_EPI_FINISH=$s python bench.py --mode train --steps 12 --warmup 4 --no-secondary --no-cpu-baseline --no-roofline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print('pretrain fin=$s', d['value'], d['ms_per_step'])"
  done
done
for s in 1 0; do
  CODEPS_EPI_FINISH=$s python bench.py --steps 30 --warmup 8 --no-secondary --no-cpu-baseline --no-roofline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print('adapt fin=$s', d['value'], d['ms_per_step'])"
done
