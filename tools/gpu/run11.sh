mkdir -p gpurun_out/r5; rm -f gpurun_out/r5/chk_*
CODEPS_EPI_CHECK=$PWD/gpurun_out/r5/chk python -m pytest tests/test_distributed_gpu.py -q -x -k "sharded_step and True" 2>&1 | grep -E "rel-L2|passed|failed" | head -3
for f in gpurun_out/r5/chk_*; do echo == $f; cat $f | head -40; done
