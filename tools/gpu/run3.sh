#!/bin/bash
mkdir -p gpurun_out/r5
export TMPDIR=/tmp
python -m pytest tests/test_round5_epilogue_gpu.py -q -s > gpurun_out/r5/t_new.log 2>&1
echo "new tests rc=$?"
grep -E "BN case|passed|failed|Error|assert" gpurun_out/r5/t_new.log | head -40
(cd _head && python -m pytest tests/test_distributed_gpu.py -q -x -k configs3 > ../gpurun_out/r5/t_dist_head.log 2>&1; echo "HEAD dist rc=$?"; tail -3 ../gpurun_out/r5/t_dist_head.log)
b() { python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-roofline 2>/dev/null | grep '^{"metric"' | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$1', d['value'], d['ms_per_step'], d.get('step_ms'))"; }
for i in 1 2; do
  CODEPS_EPILOGUE=0 b "none "
  CODEPS_EPI_ACT=0 CODEPS_EPI_BN=0 b "stats"
  CODEPS_EPI_STATS=0 CODEPS_EPI_BN=0 b "act  "
  CODEPS_EPI_STATS=0 CODEPS_EPI_ACT=0 b "bn   "
  b "all  "
done
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_e1 -o r -- python3 bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-roofline > /dev/null 2>&1
cp $(find gpurun_out/prof_e1 -name "*kernel_stats.csv" | head -1) gpurun_out/r5/stats_epi1.csv
rm -rf gpurun_out/prof_e1
