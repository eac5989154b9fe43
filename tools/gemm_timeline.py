#!/usr/bin/env python
"""Developer probe: per-workgroup phase times of the fp32 GEMM (shader-clock stamps)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from sopro_amd import hip
DEV = "cuda:0"
lib = hip.load()
for name, M, N, K, cfg, pro in [("convT4 128x128", 3072000, 256, 256, 1, hip.PRO_ELU), ("convT4 64x64", 3072000, 256, 256, 5, hip.PRO_ELU),
                                ("convT4 no-elu", 3072000, 256, 256, 1, hip.PRO_NONE), ("convT3", 614400, 640, 512, 1, hip.PRO_ELU),
                                ("head 64x64", 6400, 2048, 256, 5, hip.PRO_NONE)]:
    A = torch.randn(M, K, device=DEV); W = torch.randn(N, K, device=DEV) * K ** -0.5; b = torch.zeros(N, device=DEV)
    Cc = torch.empty(M, N, device=DEV)
    lib.sopro_gemm_set_tile_override(cfg)
    bm = 128 if cfg == 1 else 64
    nwg = ((M + bm - 1) // bm) * ((N + bm - 1) // bm)
    dbg = torch.zeros(nwg, 8, dtype=torch.int64, device=DEV)
    for _ in range(2):
        hip.gemm(A, W, Cc, M=M, N=N, K=K, bias=b, prologue=pro, dbg=dbg)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); hip.gemm(A, W, Cc, M=M, N=N, K=K, bias=b, prologue=pro); e1.record(); torch.cuda.synchronize()
    d = dbg.cpu().double()
    span = float(d[:, 3].max() - d[:, 0].min())
    print(f"{name:15s}: kernel {e0.elapsed_time(e1)*1e3:8.1f} us = {span:10.0f} cyc ({span/ (e0.elapsed_time(e1)*1e3):.0f} MHz) | per WG median: prologue {float((d[:,1]-d[:,0]).median()):7.0f}, "
          f"main loop {float((d[:,2]-d[:,1]).median()):7.0f}, epilogue {float((d[:,3]-d[:,2]).median()):7.0f}, total {float((d[:,3]-d[:,0]).median()):7.0f} cyc; WGs {nwg}")
lib.sopro_gemm_set_tile_override(0)
