#!/usr/bin/env python
"""r06: where a tile of the three-pass tile kernel spends its life on the 192-CU partition: shader-clock stamps of wave 0 at the start,
behind the prologue (first rows staged), behind the K loop, at the end - medians over the workgroups of one launch, and how many
workgroups were resident at a time (sum of lives / (span x CUs)).   python tools/r06/tile_life.py"""
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
SHAPES = (("nar glu rms", 25600, 768, 384, "f16x3", dict(epilogue=hip.EPI_GLU, rms_eps=1e-6)), ("nar heads argmax", 25600, 8192, 256, "f16x3", dict(c_mode=5)),
          ("nar ff1 rms+gelu", 25600, 1536, 384, "f16x3", dict(epilogue=hip.EPI_GELU, rms_eps=1e-6)), ("nar ff2 res", 25600, 384, 1536, "f16x3", dict(epilogue=hip.EPI_RES)),
          ("mimi qkv", 25600, 1536, 512, "bf16x3", {}), ("mimi o res", 25600, 512, 512, "bf16x3", dict(epilogue=hip.EPI_RES)),
          ("mimi fc1 gelu", 25600, 2048, 512, "bf16x3", dict(epilogue=hip.EPI_GELU)), ("mimi fc2 res", 25600, 512, 2048, "bf16x3", dict(epilogue=hip.EPI_RES)),
          ("up2", 307200, 640, 512, "bf16x3", {}))
for name, M, N, K, kind, kw in SHAPES:
    A, W, b, Rr = rn(M, K), rn(N, K, scale=K ** -0.5), rn(N, scale=0.1), rn(M, N)
    Wp = hip.pack_w_f16x3(W) if kind == "f16x3" else hip.pack_w_bf16x3(W)
    Cc = torch.empty(M, N // 2 if kw.get("epilogue") == hip.EPI_GLU else N, device=DEV) if kw.get("c_mode") != 5 else None
    kw = dict(kw, M=M, N=N, K=K, bias=b)
    if kw.get("c_mode") == 5:
        kw.update(C2=torch.zeros(M, N // 64, 2, device=DEV), ldc2=N // 64)
    if kw.get("epilogue") == hip.EPI_RES:
        kw["R"] = Rr
    nwg = ((M + 127) // 128) * ((N + 127) // 128)
    dbg = torch.zeros(nwg, 8, dtype=torch.int64, device=DEV)
    with torch.cuda.stream(st):
        for _ in range(3):
            hip.gemm(A, Wp, Cc, dbg=dbg, **kw)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(st)
        for _ in range(10):
            hip.gemm(A, Wp, Cc, **kw)
        e1.record(st)
    torch.cuda.synchronize()
    d = dbg.cpu().double()
    d = d[d[:, 0] > 0]  # (rows of workgroups that were never stamped stay zero)
    us = e0.elapsed_time(e1) * 1e2
    span = float(d[:, 3].max() - d[:, 0].min())
    life = d[:, 3] - d[:, 0]
    med = lambda x: float(x.median())  # noqa: E731
    kt = (K + 31) // 32
    mfma = kt * 24 * 32  # cycles of one wave's MFMAs in the K loop (24 per K-step, 8 passes of 4 clocks)
    print(f"{name:17s} {kind} {M} x {N} x {K}: {us:7.1f} us/launch; stamped launch spans {span:9.0f} clk; per workgroup (median clk): prologue {med(d[:, 1] - d[:, 0]):6.0f}, "
          f"K loop {med(d[:, 2] - d[:, 1]):6.0f} ({med(d[:, 2] - d[:, 1]) / kt:5.0f} per K-step; its MFMAs alone {mfma}), epilogue {med(d[:, 3] - d[:, 2]):6.0f}, "
          f"(transposition {med(d[:, 4] - d[:, 2]):6.0f}, first 8 rows per thread {med(d[:, 5] - d[:, 4]):6.0f}, last 8 {med(d[:, 3] - d[:, 5]):6.0f}: developer build only) life {med(life):6.0f}; resident at a time {float(life.sum()) / span / 192:4.2f} per CU; {nwg} workgroups", flush=True)
