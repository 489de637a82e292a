python tools/gpu/dbg_stats2.py 2>&1 | grep -v amdgpu.ids
