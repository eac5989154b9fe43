#!/usr/bin/env python
"""r06: tile shapes of the three-pass tile kernel on the decoder's / refinement's K <= 2048 contractions, 192-CU partition, HIP events over
30 launches.  Developer build (SOPRO_DEV=1): tile override 1 = 128 x 128 on 2 x 2 waves of 64 x 64 (the product's), 3 = 128 x 128 on
1 x 4 waves of 128 x 32 (every W fragment requested once per workgroup).   SOPRO_DEV=1 python tools/r06/tile_probe.py [tiles ...]"""
import os
import sys

R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, R)
import torch  # noqa: E402

from sopro_amd import hip  # noqa: E402

DEV = torch.device("cuda:0")
torch.cuda.set_device(0)
st = hip.cu_range_stream(64, 192, DEV)
g = torch.Generator(device=DEV).manual_seed(1)
rn = lambda *s, scale=1.0: torch.randn(*s, device=DEV, generator=g) * scale  # noqa: E731
TILES = [int(a) for a in sys.argv[1:]] or [1, 3]
lib = hip.load()


def timed(fn, n=30):
    with torch.cuda.stream(st):
        for _ in range(3):
            fn()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(st)
        for _ in range(n):
            fn()
        e1.record(st)
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


SHAPES = (("nar ff1 rms", 25600, 1536, 384, "f16x3", -hip.EPI_GELU), ("nar glu rms", 25600, 768, 384, "f16x3", -hip.EPI_GLU), ("nar glu", 25600, 768, 384, "f16x3", hip.EPI_GLU),
          ("nar plain rms", 25600, 256, 384, "f16x3", -1000),
          ("nar ff1", 25600, 1536, 384, "f16x3", hip.EPI_GELU), ("nar glu-like", 25600, 768, 384, "f16x3", 0), ("nar ff2", 25600, 384, 1536, "f16x3", hip.EPI_RES),
          ("mimi qkv", 25600, 1536, 512, "bf16x3", 0), ("mimi o", 25600, 512, 512, "bf16x3", hip.EPI_RES), ("mimi fc1", 25600, 2048, 512, "bf16x3", hip.EPI_GELU),
          ("mimi fc2", 25600, 512, 2048, "bf16x3", hip.EPI_RES), ("up2", 307200, 640, 512, "bf16x3", 0), ("res1.c1", 204800, 128, 768, "bf16x3", 0))
for name, M, N, K, kind, epi in SHAPES:
    A, W, b, Rr = rn(M, K), rn(N, K, scale=K ** -0.5), rn(N, scale=0.1), rn(M, N)
    Wp = hip.pack_w_f16x3(W) if kind == "f16x3" else hip.pack_w_bf16x3(W)
    outs, row = [], [f"{name:13s} {kind:7s} {M} x {N} x {K}:"]
    for t in TILES:
        lib.sopro_gemm_bf16_set_tile_override(t)
        e = abs(epi) if epi != -1000 else 0
        Cc = torch.empty(M, N // 2 if e == hip.EPI_GLU else N, device=DEV)
        kw = dict(M=M, N=N, K=K, bias=b)
        if epi < 0:  # fused RMSNorm on the rows
            kw["rms_eps"] = 1e-6
        if e:
            kw["epilogue"] = e
        if e == hip.EPI_RES:
            kw["R"] = Rr
        us = timed(lambda: hip.gemm(A, Wp, Cc, **kw))
        outs.append(Cc)
        row.append(f"tile {t}: {us:7.1f} us ({2e-6 * M * N * K / us:6.1f} TFLOP/s fp32-eq)")
        torch.cuda.synchronize()
    lib.sopro_gemm_bf16_set_tile_override(0)
    same = all(torch.equal(outs[0], o) for o in outs[1:])
    print("  ".join(row) + f"  same bits: {same}", flush=True)
