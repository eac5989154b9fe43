#!/usr/bin/env python
"""Developer probe: time the fp32 GEMM on the shapes of the NAR / Mimi phases under each tile shape."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from sopro_amd import hip

DEV = "cuda:0"
shapes = [  # (name, M, N, K, epilogue)
    ("nar.glu", 6400, 768, 384, hip.EPI_GLU), ("nar.ff1", 6400, 1536, 384, hip.EPI_GELU), ("nar.ff2", 6400, 384, 1536, hip.EPI_RES),
    ("nar.head", 6400, 2048, 256, hip.EPI_NONE), ("tr.qkv", 12800, 1536, 512, hip.EPI_NONE), ("tr.o", 12800, 512, 512, hip.EPI_RES),
    ("tr.fc1", 12800, 2048, 512, hip.EPI_GELU), ("tr.fc2", 12800, 512, 2048, hip.EPI_RES), ("convT4", 3072000, 256, 256, hip.EPI_NONE),
    ("res3.c1", 3072000, 64, 384, hip.EPI_NONE), ("res3.c2", 3072000, 128, 64, hip.EPI_RES), ("res2.c2", 614400, 256, 128, hip.EPI_RES),
]
lib = hip.load()
for name, M, N, K, epi in shapes:
    A = torch.randn(M, K, device=DEV)
    W = torch.randn(N, K, device=DEV) * K ** -0.5
    b = torch.zeros(N, device=DEV)
    nout = N // 2 if epi == hip.EPI_GLU else N
    Cc = torch.empty(M, nout, device=DEV)
    R = torch.randn(M, nout, device=DEV) if epi == hip.EPI_RES else None
    res = []
    for cfg in (1, 2, 3, 5):
        if epi == hip.EPI_GLU and cfg == 5:
            continue
        lib.sopro_gemm_set_tile_override(cfg)
        for _ in range(2):
            hip.gemm(A, W, Cc, M=M, N=N, K=K, bias=b, epilogue=epi, R=R)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(5):
            hip.gemm(A, W, Cc, M=M, N=N, K=K, bias=b, epilogue=epi, R=R)
        e1.record()
        torch.cuda.synchronize()
        us = e0.elapsed_time(e1) / 5 * 1e3
        res.append(f"cfg{cfg}: {us:8.1f} us {2.0 * M * N * K / us / 1e6:6.1f} TF")
    lib.sopro_gemm_set_tile_override(0)
    print(f"{name:9s} M={M:8d} N={N:5d} K={K:5d} | " + " | ".join(res))
