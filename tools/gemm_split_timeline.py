#!/usr/bin/env python
"""Developer probe: per-workgroup phase times of the split-bf16 GEMM (shader-clock stamps)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from sopro_amd import hip
DEV = "cuda:0"
lib = hip.load()
EL = hip.PRO_ELU
for name, M, N, K, cfg, pro, kw in [("up1 128x128 elu", 102400, 1536, 1024, 1, EL, {}), ("up1 128x128 plain", 102400, 1536, 1024, 1, 0, {}),
                                    ("up1 split-in", 102400, 1536, 1024, 1, 0, dict(a_split=True)),
                                    ("up1 split-in dual", 102400, 1536, 1024, 1, 0, dict(a_split=True, c_mode=2)),
                                    ("up1 plain dual", 102400, 1536, 1024, 1, 0, dict(c_mode=2)),
                                    ("up3 128x128 elu", 3072000, 256, 256, 1, EL, {}), ("up3 split-in", 3072000, 256, 256, 1, 0, dict(a_split=True))]:
    A = torch.randn(M, K, device=DEV); W = torch.randn(N, K, device=DEV) * K ** -0.5; b = torch.zeros(N, device=DEV)
    Wp = hip.pack_w_bf16x3(W)
    Cc = torch.empty(M, N, device=DEV)
    if kw.get("a_split"):  # a well-formed split-form operand: ELU(A) as [32 hi | 32 lo] per 32 channels
        e = torch.nn.functional.elu(A).view(M, K // 32, 32)
        hi = e.bfloat16()
        lo = (e - hi.float()).bfloat16()
        A = torch.stack([hi, lo], dim=2).reshape(M, K * 2).contiguous().view(torch.float32).view(M, K)
    if kw.get("c_mode") == 2:
        kw = dict(kw, C2=torch.empty(M, N, device=DEV))
    lib.sopro_gemm_bf16_set_tile_override(cfg)
    bm, bn = {1: (128, 128), 2: (256, 128), 4: (128, 64), 5: (64, 64)}[cfg]
    nwg = ((M + bm - 1) // bm) * ((N + bn - 1) // bn)
    dbg = torch.zeros(nwg, 8, dtype=torch.int64, device=DEV)
    for _ in range(2):
        hip.gemm(A, Wp, Cc, M=M, N=N, K=K, bias=b, prologue=pro, dbg=dbg, **kw)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); hip.gemm(A, Wp, Cc, M=M, N=N, K=K, bias=b, prologue=pro, **kw); e1.record(); torch.cuda.synchronize()
    d = dbg.cpu().double()
    span = float(d[:, 3].max() - d[:, 0].min())
    us = e0.elapsed_time(e1) * 1e3
    kt = K // 32
    ml = float((d[:, 2] - d[:, 1]).median())
    print(f"{name:18s}: {us:8.1f} us {2.0*M*N*K/us/1e6:6.1f} TF, clock {span/us:.0f} MHz | per WG median: prologue {float((d[:,1]-d[:,0]).median()):7.0f}, "
          f"main loop {ml:7.0f} ({ml/kt:6.0f}/K-step), epilogue {float((d[:,3]-d[:,2]).median()):7.0f}, total {float((d[:,3]-d[:,0]).median()):7.0f} cyc; WGs {nwg}")
lib.sopro_gemm_bf16_set_tile_override(0)
