#!/usr/bin/env python
"""Developer probe: a few large decoder contractions on the three-pass split kernel, timed back to back - run once per build
(SOPRO_HIP_LIB=tools/micro/libsopro_ablN.so for an ablation build) to A/B a change of the K loop.  Prints a checksum per shape:
builds that claim the same arithmetic must print the same sums."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from sopro_amd import hip

DEV = "cuda:0"
shapes = [("sea.conv0", 12800, 1024, 3584), ("up0", 12800, 4096, 2048), ("up1", 102400, 1536, 1024), ("tr.fc1", 25600, 2048, 512), ("up2", 614400, 640, 512)]
reps = int(os.environ.get("PROBE_REPS", "6"))
out = []
for name, M, N, K in shapes:
    g = torch.Generator(device=DEV).manual_seed(1)
    A = torch.randn(M, K, device=DEV, generator=g)
    W = torch.randn(N, K, device=DEV, generator=g) * K ** -0.5
    Wp = hip.pack_w_bf16x3(W)
    C = torch.empty(M, N, device=DEV)
    for _ in range(2):
        hip.gemm(A, Wp, C, M=M, N=N, K=K)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        hip.gemm(A, Wp, C, M=M, N=N, K=K)
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / reps * 1e3
    out.append(f"{name} {us:7.1f}us {2.0 * M * N * K / us / 1e6:5.1f}TF sum={float(C.double().sum()):.6e}")
print(os.environ.get("SOPRO_HIP_LIB", "product build"), "|", " | ".join(out), flush=True)
