#!/bin/bash
mkdir -p gpurun_out/r5
python -m pytest tests -x -q -m gpu > gpurun_out/r5/t42.log 2>&1
echo "all tests rc=$?"; tail -3 gpurun_out/r5/t42.log
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -1
