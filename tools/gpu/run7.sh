python tools/gpu/dbg_bn.py 6,32,64,256,512,2,2,2 6,32,64,64,64,2,1,3 2>&1 | grep -v amdgpu.ids
