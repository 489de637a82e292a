#!/bin/bash
b() { python bench.py --mode train --steps 10 --warmup 3 --no-cpu-baseline --no-roofline --no-secondary 2>/dev/null | grep '^{"metric"' | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$1', d['value'], d['ms_per_step'], d.get('step_ms'))"; }
for i in 1 2; do
  CODEPS_EPILOGUE=0 b "none "
  CODEPS_EPI_ACT=0 CODEPS_EPI_BN=0 b "stats"
  CODEPS_EPI_STATS=0 CODEPS_EPI_BN=0 b "act  "
  CODEPS_EPI_STATS=0 CODEPS_EPI_ACT=0 b "bn   "
  b "all  "
done
