#!/usr/bin/env python
"""Developer probe: time of the split contractions (three-pass bf16x3 and one-pass bf16x1, 128x128 tiles) on a few decoder /
refinement shapes - meant to be run once with the product library and once with a timing build (SOPRO_HIP_LIB=...: e.g. the K loop
without its workgroup barriers: round 4's patch, removed in round 5 - last tree 9bb62d2, numbers in profiles/r04_gemm_no_barriers.txt)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from sopro_amd import hip

DEV = "cuda:0"
shapes = [("up0", 12800, 4096, 2048), ("tr.qkv", 12800, 1536, 512), ("tr.fc2", 12800, 512, 2048), ("up1", 102400, 1536, 1024), ("up2", 614400, 640, 512),
          ("nar.ff1", 12800, 1536, 384), ("nar.ff2", 12800, 384, 1536)]
lib = hip.load()
if os.environ.get("PROBE_CUS"):
    n_cus = int(os.environ["PROBE_CUS"])
    torch.cuda.set_stream(hip.cu_range_stream(256 - n_cus, n_cus, torch.device(DEV)))
tag = os.path.basename(os.environ.get("SOPRO_HIP_LIB", "product"))
for name, M, N, K in shapes:
    g = torch.Generator(device=DEV).manual_seed(1)
    A = torch.randn(M, K, device=DEV, generator=g)
    W = torch.randn(N, K, device=DEV, generator=g) * K ** -0.5
    C = torch.empty(M, N, device=DEV)
    out = []
    for pieces, Wp in ((2, hip.pack_w_bf16x3(W)), (1, hip.pack_w_bf16x1(W))):
        lib.sopro_gemm_bf16_set_tile_override(1 if pieces == 2 else 0)
        for _ in range(2):
            hip.gemm(A, Wp, C, M=M, N=N, K=K)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(5):
            hip.gemm(A, Wp, C, M=M, N=N, K=K)
        e1.record()
        torch.cuda.synchronize()
        us = e0.elapsed_time(e1) / 5 * 1e3
        out.append(f"{'bf16x3' if pieces == 2 else 'bf16x1'} {us:8.1f} us {2.0 * M * N * K / us / 1e6:6.1f} TF")
    lib.sopro_gemm_bf16_set_tile_override(0)
    print(f"[{tag}] {name:8s} M={M:7d} N={N:5d} K={K:5d} | " + " | ".join(out), flush=True)
