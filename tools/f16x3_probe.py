#!/usr/bin/env python
"""Developer probe: error of the f16x3 contraction against fp64 by activation scale (does the matrix core flush fp16 subnormal
operands?).  Mean / max of |err| / (|A||W|) for rows of RMS s, for s over six decades, next to bf16x6 and bf16x3."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from sopro_amd import hip

DEV = "cuda:0"
torch.manual_seed(0)
M, N, K = 512, 256, 384
A0 = torch.randn(M, K)
W = torch.randn(N, K) * K ** -0.5
packs = {"f16x3": hip.pack_w_f16x3(W.to(DEV)), "bf16x6": hip.pack_w_bf16x6(W.to(DEV)), "bf16x3": hip.pack_w_bf16x3(W.to(DEV))}
for s in (1e-3, 1e-2, 0.1, 1.0, 10.0, 100.0, 1000.0):
    A = A0 * s
    ref = A.double() @ W.double().t()
    aw = A.double().abs() @ W.double().abs().t()
    line = f"row RMS {s:8.3g}:"
    for name, Wp in packs.items():
        C = torch.empty(M, N, device=DEV)
        hip.gemm(A.to(DEV), Wp, C, M=M, N=N, K=K)
        torch.cuda.synchronize()
        e = (C.cpu().double() - ref).abs() / aw
        line += f"  {name}: mean {float(e.mean()):.2e} max {float(e.max()):.2e}"
    print(line, flush=True)
