python tools/f16x3_probe.py 2>&1 | grep -v amdgpu.ids
