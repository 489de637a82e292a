python tools/gpu/dbg_stats.py 2>&1 | grep -v amdgpu.ids
