#!/bin/bash
# first GPU pass of round 5: the epilogue tests, the whole GPU suite, then an A/B of the step with / without the epilogue extras
mkdir -p gpurun_out/r5
python -m pytest tests/test_round5_epilogue_gpu.py -x -q > gpurun_out/r5/t_new.log 2>&1
echo "new tests rc=$?" 
tail -25 gpurun_out/r5/t_new.log
python -m pytest tests -m gpu -q -x --deselect tests/test_round5_epilogue_gpu.py > gpurun_out/r5/t_all.log 2>&1
echo "all tests rc=$?"
tail -15 gpurun_out/r5/t_all.log
for i in 1 2; do
  CODEPS_EPILOGUE=0 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-roofline 2>/dev/null | grep '^{"metric"' | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('EPI=0', d['value'], d['ms_per_step'], d.get('step_ms'))"
  CODEPS_EPILOGUE=1 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-roofline 2>/dev/null | grep '^{"metric"' | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('EPI=1', d['value'], d['ms_per_step'], d.get('step_ms'))"
done
