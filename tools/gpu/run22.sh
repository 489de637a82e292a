#!/bin/bash
mkdir -p gpurun_out/r5
python -m pytest tests -m gpu -q -x > gpurun_out/r5/t_all.log 2>&1
echo "all tests rc=$?"; tail -4 gpurun_out/r5/t_all.log
b() { python bench.py $2 --steps 20 --warmup 5 --no-cpu-baseline --no-roofline --no-secondary 2>/dev/null | grep '^{"metric"' | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$1', d['value'], d['ms_per_step'], d.get('step_ms'))"; }
for i in 1 2; do
  CODEPS_PARITY_MERGE=0 b "adapt merge0" ""
  b "adapt merge1" ""
  CODEPS_EPILOGUE=0 b "adapt epi0  " ""
done
b "pretrain    " "--mode train"
CODEPS_EPILOGUE=0 b "pretrain epi0" "--mode train"
CODEPS_EPI_BN=0 b "pretrain bn0 " "--mode train"
b "pretrain    " "--mode train"
CODEPS_EPILOGUE=0 b "pretrain epi0" "--mode train"
CODEPS_EPI_BN=0 b "pretrain bn0 " "--mode train"
