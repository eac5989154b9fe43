#!/usr/bin/env python
"""r06: the fused-RMSNorm forms of the f16 three-pass kernel on the three tile shapes the engine uses (override 1 = 128 x 128, 4 = 64 x 128,
5 = 64 x 64 / 64 x 128 for GLU): do the bits depend on the tile shape?   python tools/r06/form_tiles.py"""
import hashlib
import os
import sys

R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, R)
import torch  # noqa: E402

from sopro_amd import hip  # noqa: E402

DEV = torch.device("cuda:0")
torch.cuda.set_device(0)
lib = hip.load()
g = torch.Generator(device=DEV).manual_seed(7)
rn = lambda *s, scale=1.0: torch.randn(*s, device=DEV, generator=g) * scale  # noqa: E731
h = lambda t: hashlib.sha1(t.detach().cpu().contiguous().view(torch.uint8).numpy().tobytes()).hexdigest()[:12]  # noqa: E731
M, K = 4136, 384
for name, N, kw in (("rms plain", 256, {}), ("rms gelu", 1536, dict(epilogue=hip.EPI_GELU)), ("rms glu", 768, dict(epilogue=hip.EPI_GLU)), ("glu", 768, dict(epilogue=hip.EPI_GLU, rms_eps=0.0))):
    A, W, b = rn(M, K), rn(N, K, scale=K ** -0.5), rn(N, scale=0.1)
    Wp = hip.pack_w_f16x3(W)
    row = []
    for t in (1, 4, 5):
        lib.sopro_gemm_bf16_set_tile_override(t)
        out = torch.full((M, N // 2 if kw.get("epilogue") == hip.EPI_GLU else N), float("nan"), device=DEV)
        hip.gemm(A, Wp, out, M=M, N=N, K=K, bias=b, **dict(dict(rms_eps=1e-6), **kw))
        torch.cuda.synchronize()
        row.append(f"tile {t}: {h(out)}")
    lib.sopro_gemm_bf16_set_tile_override(0)
    print(f"{name:10s} {M} x {N} x {K}:  " + "  ".join(row), flush=True)
