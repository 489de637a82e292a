#!/bin/bash
mkdir -p gpurun_out/r5
python -m pytest tests -m gpu -q > gpurun_out/r5/t_all.log 2>&1
echo "all tests rc=$?"
tail -4 gpurun_out/r5/t_all.log
python bench.py --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | grep '^{"metric"' | tail -1 > gpurun_out/r5/bench_a.json
python -c "import json; d=json.load(open('gpurun_out/r5/bench_a.json')); print(d['value'], d['ms_per_step'], d['roofline'])"
