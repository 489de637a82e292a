#!/bin/bash
mkdir -p gpurun_out/r5
export TMPDIR=/tmp
python -m pytest tests/test_round5_epilogue_gpu.py -q -s > gpurun_out/r5/t_new.log 2>&1
echo "new tests rc=$?"
grep -E "BN case|gradient|passed|failed|Error|assert" gpurun_out/r5/t_new.log | head -80
python -m pytest tests/test_distributed_gpu.py -q -x > gpurun_out/r5/t_dist.log 2>&1
echo "dist rc=$?"; tail -5 gpurun_out/r5/t_dist.log
CODEPS_EPILOGUE=0 python -m pytest tests/test_distributed_gpu.py -q -x > gpurun_out/r5/t_dist0.log 2>&1
echo "dist (EPI=0) rc=$?"; tail -3 gpurun_out/r5/t_dist0.log
for e in 0 1; do
  CODEPS_EPILOGUE=$e rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_e$e -o r -- python3 bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-roofline > /dev/null 2>&1
  cp $(find gpurun_out/prof_e$e -name "*kernel_stats.csv" | head -1) gpurun_out/r5/stats_epi$e.csv
  rm -rf gpurun_out/prof_e$e
done
for i in 1 2; do
  for e in 0 1; do
  CODEPS_EPILOGUE=$e python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-roofline 2>/dev/null | grep '^{"metric"' | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('EPI=$e', d['value'], d['ms_per_step'], d.get('step_ms'))"
  done
done
