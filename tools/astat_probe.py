#!/usr/bin/env python
"""Developer probe (round 5): the activation-stationary short-K contraction (csrc/gemm_astat.hip, tile override 9) against the 128x128
tile kernel (override 1) on the decoder's / refinement's K <= 512 shapes.  PROBE_CUS=192: on the throughput partition's CUs."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from sopro_amd import hip

DEV = "cuda:0"
shapes = [("tr.qkv", 25600, 1536, 512, 0), ("tr.fc1", 25600, 2048, 512, 1), ("tr.o", 25600, 512, 512, 3), ("up2", 1228800, 640, 512, 0),
          ("rvq", 12800, 512, 512, 0), ("nar.ff1*", 12800, 1536, 384, 1), ("nar.glu*", 12800, 768, 384, 0), ("head*", 12800, 8192, 256, 0)]
lib = hip.load()
if os.environ.get("PROBE_CUS"):
    n_cus = int(os.environ["PROBE_CUS"])
    torch.cuda.set_stream(hip.cu_range_stream(256 - n_cus, n_cus, torch.device(DEV)))
for name, M, N, K, epi in shapes:
    g = torch.Generator(device=DEV).manual_seed(1)
    A = torch.randn(M, K, device=DEV, generator=g)
    W = torch.randn(N, K, device=DEV, generator=g) * K ** -0.5
    b = torch.randn(N, device=DEV, generator=g)
    C = torch.empty(M, N, device=DEV)
    R = torch.randn(M, N, device=DEV, generator=g) if epi == 3 else None
    Wp = hip.pack_w_bf16x3(W)
    out, res = [], []
    for cfg in (1, 9):
        lib.sopro_gemm_bf16_set_tile_override(cfg)
        kw = dict(M=M, N=N, K=K, bias=b, epilogue=epi)
        if epi == 3:
            kw["R"] = R
        for _ in range(2):
            hip.gemm(A, Wp, C, **kw)
        res.append(C.clone())
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(5):
            hip.gemm(A, Wp, C, **kw)
        e1.record()
        torch.cuda.synchronize()
        us = e0.elapsed_time(e1) / 5 * 1e3
        out.append(f"{'tile ' if cfg == 1 else 'astat'} {us:8.1f} us {2.0 * M * N * K / us / 1e6:6.1f} TF")
    lib.sopro_gemm_bf16_set_tile_override(0)
    print(f"{name:9s} M={M:8d} N={N:5d} K={K:4d} epi={epi} | " + " | ".join(out) + f" | identical {bool(torch.equal(res[0], res[1]))}", flush=True)
