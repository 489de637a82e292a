#!/bin/bash
mkdir -p gpurun_out/r5
python -m pytest tests/test_round5_epilogue_gpu.py -x -q -m gpu -k "16_channel or decoder_tail or activation_backward" > gpurun_out/r5/t29.log 2>&1
echo "tests rc=$?"; tail -15 gpurun_out/r5/t29.log
for i in 1 2; do
for s in 1 0; do
CODEPS_HALO_N16=$s python bench.py --steps 30 --warmup 8 --no-secondary --no-cpu-baseline --no-roofline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print('adapt n16=$s', d['value'], d['ms_per_step'])"
done
done
