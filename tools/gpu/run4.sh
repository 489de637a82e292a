#!/bin/bash
mkdir -p gpurun_out/r5
(cd _head && python -m pytest tests/test_distributed_gpu.py -q -x -k configs3 > ../gpurun_out/r5/t_dist_head_newso.log 2>&1; echo "HEAD python + new .so: dist rc=$?"; tail -3 ../gpurun_out/r5/t_dist_head_newso.log)
