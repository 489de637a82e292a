#!/bin/bash
for v in "CODEPS_EPI_ACT=0" "CODEPS_EPI_BN=0" "CODEPS_EPI_STATS=0" "CODEPS_EPILOGUE=0"; do
  echo "== $v"; env $v python -m pytest tests/test_distributed_gpu.py -q -x -k "sharded_step and True" 2>&1 | grep -E "rel-L2|passed|failed" | head -3
done
