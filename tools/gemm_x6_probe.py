#!/usr/bin/env python
"""Developer probe: the six-pass split-bf16 GEMM on the NAR shapes: accuracy against fp64 and time per tile shape, with
the fp32-MFMA kernel beside it."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from sopro_amd import hip, pack

DEV = "cuda:0"
shapes = [("nar.glu", 6400, 768, 384, hip.EPI_GLU, 0), ("nar.ff1", 6400, 1536, 384, hip.EPI_GELU, 0), ("nar.ff2", 6400, 384, 1536, hip.EPI_RES, 0),
          ("nar.head", 6400, 2048, 256, hip.EPI_NONE, hip.PRO_ADDVEC), ("nar.pre", 6400, 256, 384, hip.EPI_NONE, 0),
          ("txt.ff1", 2048, 1536, 384, hip.EPI_GELU, 0)]
lib = hip.load()
for name, M, N, K, epi, pro in shapes:
    g = torch.Generator(device=DEV).manual_seed(1)
    A = torch.randn(M, K, device=DEV, generator=g)
    W = torch.randn(N, K, device=DEV, generator=g) * K ** -0.5
    b = torch.randn(N, device=DEV, generator=g)
    pv = torch.randn(K, device=DEV, generator=g) if pro else None
    if epi == hip.EPI_GLU:
        Wk, bk = pack.pack_glu(W.cpu(), b.cpu())
        Wk, bk = Wk.to(DEV), bk.to(DEV)
    else:
        Wk, bk = W, b
    Wp = hip.pack_w_bf16x6(Wk)
    nout = N // 2 if epi == hip.EPI_GLU else N
    Cc = torch.empty(M, nout, device=DEV)
    R = torch.randn(M, nout, device=DEV, generator=g) if epi == hip.EPI_RES else None
    a64 = A[:256].double() + (pv.double() if pro else 0)
    pre = a64 @ W.double().t() + b.double()
    mag = a64.abs() @ W.double().abs().t() + b.double().abs()
    if epi == hip.EPI_GELU:
        ref = torch.nn.functional.gelu(pre)
    elif epi == hip.EPI_RES:
        ref = pre + R[:256].double()
    elif epi == hip.EPI_GLU:
        ref, mag = pre[:, : N // 2] * torch.sigmoid(pre[:, N // 2:]), mag[:, : N // 2]
    else:
        ref = pre
    kw = dict(bias=bk, epilogue=epi, prologue=pro, R=R, pro_vec=pv)
    res = []
    for cfg in (1, 4, 5):
        if epi == hip.EPI_GLU and cfg == 5:
            continue
        lib.sopro_gemm_bf16_set_tile_override(cfg)
        for _ in range(2):
            hip.gemm(A, Wp, Cc, M=M, N=N, K=K, **kw)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10):
            hip.gemm(A, Wp, Cc, M=M, N=N, K=K, **kw)
        e1.record()
        torch.cuda.synchronize()
        us = e0.elapsed_time(e1) / 10 * 1e3
        err = float(((Cc[:256].double() - ref).abs() / mag).max())
        res.append(f"c{cfg}:{us:7.1f}us {2.0 * M * N * K / us / 1e6:6.1f}TF e={err:.1e}")
    lib.sopro_gemm_bf16_set_tile_override(0)
    for _ in range(2):
        hip.gemm(A, Wk, Cc, M=M, N=N, K=K, **kw)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10):
        hip.gemm(A, Wk, Cc, M=M, N=N, K=K, **kw)
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / 10 * 1e3
    e32 = float(((Cc[:256].double() - ref).abs() / mag).max())
    print(f"{name:9s} M={M:6d} N={N:5d} K={K:5d} | " + " | ".join(res) + f" | f32:{us:7.1f}us {2.0 * M * N * K / us / 1e6:6.1f}TF e={e32:.1e}", flush=True)
