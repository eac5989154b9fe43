"""Registers / spills / occupancy of the kernels of one csrc file (hipcc -Rpass-analysis=kernel-resource-usage), one line per kernel.
Usage: python tools/kernel_resources.py sopro_amd/csrc/gemm_bf16s.hip [name-substring]   (build container: no GPU needed)"""
import re, subprocess, sys, os
src = sys.argv[1]; pat = sys.argv[2] if len(sys.argv) > 2 else ""
d = os.path.dirname(os.path.abspath(src))
r = subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-fno-gpu-rdc", "-c", os.path.basename(src), "-o", "/dev/null",
                    "-Rpass-analysis=kernel-resource-usage"] + sys.argv[3:], cwd=d, capture_output=True, text=True)
blocks = r.stderr.split("Function Name: ")[1:]
g = lambda b, k: int(re.search(re.escape(k) + r": (\d+)", b).group(1))
for b in blocks:
    name = b.split()[0]
    try:
        name = subprocess.run(["/opt/rocm/lib/llvm/bin/llvm-cxxfilt", name], capture_output=True, text=True).stdout.strip()
    except Exception:
        pass
    name = re.sub(r"\(anonymous namespace\)::", "", name).split("(")[0]
    if pat not in name:
        continue
    print(f"{name[:110]:110s} vgpr {g(b,' VGPRs'):3d} agpr {g(b,'AGPRs'):3d} spill {g(b,'VGPRs Spill'):3d} scratch {g(b,'ScratchSize [bytes/lane]'):4d} occ {g(b,'Occupancy [waves/SIMD]')} lds {g(b,'LDS Size [bytes/block]')}")
