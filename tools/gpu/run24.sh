#!/bin/bash
mkdir -p gpurun_out/r5
python -m pytest tests/test_round5_epilogue_gpu.py -x -q -m gpu 2>&1 | tail -2
python tools/gpu/dbg_bn_cover.py train > gpurun_out/r5/bn_cover.txt 2>&1
tail -60 gpurun_out/r5/bn_cover.txt
for i in 1 2; do
  for s in 1 0; do
    CODEPS_EPI_FINISH=$s python bench.py --mode train --steps 12 --warmup 4 --no-secondary --no-cpu-baseline --no-roofline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print('pretrain fin=$s', d['value'], d['ms_per_step'])"
  done
done
