#!/usr/bin/env python
"""Developer probe: the split-bf16 GEMM on the Mimi decoder shapes: accuracy against fp64 and time per tile shape,
with the fp32-MFMA kernel beside it."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from sopro_amd import hip

DEV = "cuda:0"
E, R_, G, EL = hip.EPI_NONE, hip.EPI_RES, hip.EPI_GELU, hip.PRO_ELU
shapes = [  # (name, M, N, K, epilogue, prologue)
    ("up0", 12800, 4096, 2048, E, EL),
    ("sea.conv0", 12800, 1024, 3584, E, 0), ("tr.qkv", 12800, 1536, 512, E, 0), ("tr.o", 12800, 512, 512, R_, 0),
    ("tr.fc1", 12800, 2048, 512, G, 0), ("tr.fc2", 12800, 512, 2048, R_, 0),
    ("up0", 12800, 4096, 2048, E, EL), ("res0.c1", 102400, 256, 1536, E, EL), ("res0.c2", 102400, 512, 256, R_, EL),
    ("up1", 102400, 1536, 1024, E, EL), ("res1.c1", 614400, 128, 768, E, EL), ("res1.c2", 614400, 256, 128, R_, EL),
    ("up2", 614400, 640, 512, E, EL), ("res2.c1", 3072000, 64, 384, E, EL), ("res2.c2", 3072000, 128, 64, R_, EL),
    ("up3", 3072000, 256, 256, E, EL),
]
if os.environ.get("PROBE_NOELU", "0") == "1":  # the activated-copy flow: the producer applied ELU, the consumer reads fp32 rows as they are
    shapes = [(n, M, N, K, e, 0) for (n, M, N, K, e, p) in shapes]
cfgs = [int(c) for c in sys.argv[1].split(",")] if len(sys.argv) > 1 else [1, 2, 3, 4, 5]
lib = hip.load()
if os.environ.get("PROBE_CUS"):  # time on a CU-masked stream (the throughput partition of the pipeline: PROBE_CUS=192)
    n_cus = int(os.environ["PROBE_CUS"])
    torch.cuda.set_stream(hip.cu_range_stream(256 - n_cus, n_cus, torch.device(DEV)))
for name, M, N, K, epi, pro in shapes:
    g = torch.Generator(device=DEV).manual_seed(1)
    A = torch.randn(M, K, device=DEV, generator=g)
    W = torch.randn(N, K, device=DEV, generator=g) * K ** -0.5
    b = torch.randn(N, device=DEV, generator=g)
    Wp = hip.pack_w_bf16x3(W)
    Cc = torch.empty(M, N, device=DEV)
    R = torch.randn(M, N, device=DEV, generator=g) if epi == R_ else None
    # accuracy on the first 256 rows against fp64
    a64 = A[:256].double()
    if pro == EL:
        a64 = torch.nn.functional.elu(a64)
    ref = a64 @ W.double().t() + b.double()
    mag = a64.abs() @ W.double().abs().t() + b.double().abs()
    if epi == G:
        ref = torch.nn.functional.gelu(ref)
    if epi == R_:
        ref = ref + R[:256].double()
    res = []
    split = os.environ.get("PROBE_SPLIT", "0") == "1" and pro == EL  # time the split-plane form of the SEANet layers
    kw = dict(bias=b, epilogue=epi, prologue=pro, R=R)
    if split:
        C2 = torch.empty(M, N, device=DEV)
        kw = dict(bias=b, epilogue=epi, R=R, a_split=True, c_mode=(2 if name.startswith("up") and name != "up3" else (0 if name == "up3" else 1)), C2=C2)
    for cfg in cfgs:
        if cfg == 3 and N < 64:
            continue
        lib.sopro_gemm_bf16_set_tile_override(cfg)
        for _ in range(2):
            hip.gemm(A, Wp, Cc, M=M, N=N, K=K, **kw)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(5):
            hip.gemm(A, Wp, Cc, M=M, N=N, K=K, **kw)
        e1.record()
        torch.cuda.synchronize()
        us = e0.elapsed_time(e1) / 5 * 1e3
        err = float("nan") if split else float(((Cc[:256].double() - ref).abs() / mag).max())
        res.append(f"c{cfg}:{us:8.1f}us {2.0 * M * N * K / us / 1e6:6.1f}TF e={err:.1e}")
    lib.sopro_gemm_bf16_set_tile_override(0)
    for _ in range(2):
        hip.gemm(A, W, Cc, M=M, N=N, K=K, bias=b, epilogue=epi, prologue=pro, R=R)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(3):
        hip.gemm(A, W, Cc, M=M, N=N, K=K, bias=b, epilogue=epi, prologue=pro, R=R)
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / 3 * 1e3
    e32 = float(((Cc[:256].double() - ref).abs() / mag).max())
    print(f"{name:9s} M={M:8d} N={N:5d} K={K:5d} | " + " | ".join(res) + f" | f32:{us:8.1f}us {2.0 * M * N * K / us / 1e6:6.1f}TF e={e32:.1e}", flush=True)
