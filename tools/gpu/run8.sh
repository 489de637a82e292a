#!/bin/bash
mkdir -p gpurun_out/r5
python -m pytest tests -m gpu -q > gpurun_out/r5/t_all.log 2>&1
echo "all tests rc=$?"
tail -12 gpurun_out/r5/t_all.log
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -3
