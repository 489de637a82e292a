#!/bin/bash
# round 5: stem statistics epilogue -- tests + A/B
mkdir -p gpurun_out/r5
python -m pytest tests/test_round5_epilogue_gpu.py tests/test_round4_ops_gpu.py tests/test_distributed_gpu.py -x -q -m gpu > gpurun_out/r5/t20.log 2>&1
echo "tests rc=$?"; tail -5 gpurun_out/r5/t20.log
for i in 1 2; do
  for s in 1 0; do
    CODEPS_STEM_STATS=$s python bench.py --steps 30 --warmup 8 --no-secondary 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print('stem_stats=$s', d['value'], d['ms_per_step'])"
  done
done
