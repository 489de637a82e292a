#!/bin/bash
mkdir -p gpurun_out/r5
python -m pytest tests -x -q -m gpu > gpurun_out/r5/t31.log 2>&1
echo "all tests rc=$?"; tail -4 gpurun_out/r5/t31.log
python bench.py --steps 30 --warmup 8 --no-secondary --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print('adapt', d['value'], d['ms_per_step'], d['roofline']['frac'], d['roofline']['conv_ms_per_step'], {k:(v['launches_per_step'], v['tflops'], v['ms_per_step']) for k,v in d['roofline']['by_kind'].items()})"
