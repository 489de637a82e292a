python tools/gpu/dbg_bn.py 6,32,64,64,64,2,1,3 2>&1 | grep -v amdgpu.ids
python tools/gpu/dbg_bn.py 4,64,128,64,128,2,2,2 2>&1 | grep -v amdgpu.ids
python -m pytest tests/test_distributed_gpu.py -q -x -k configs3 2>&1 | tail -2
