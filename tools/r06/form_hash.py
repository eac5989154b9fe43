#!/usr/bin/env python
"""r06: a bit hash of every form of the f16 / bf16 three-pass tile kernel the engine issues, on seeded operands - run under two libraries
(SOPRO_HIP_LIB) and compare the lines: which forms a change left bit-identical.   python tools/r06/form_hash.py"""
import hashlib
import os
import sys

R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, R)
import torch  # noqa: E402

from sopro_amd import hip  # noqa: E402

DEV = torch.device("cuda:0")
torch.cuda.set_device(0)
g = torch.Generator(device=DEV).manual_seed(7)
rn = lambda *s, scale=1.0: torch.randn(*s, device=DEV, generator=g) * scale  # noqa: E731
h = lambda t: hashlib.sha1(t.detach().cpu().contiguous().view(torch.uint8).numpy().tobytes()).hexdigest()[:12]  # noqa: E731
M = 4096 + 40
for name, N, K, kind, kw in (("f16 plain", 384, 1536, "f16", {}), ("f16 gelu", 1536, 384, "f16", dict(epilogue=hip.EPI_GELU)), ("f16 res", 384, 1536, "f16", dict(epilogue=hip.EPI_RES)),
                             ("f16 glu", 768, 384, "f16", dict(epilogue=hip.EPI_GLU)), ("f16 rms plain", 256, 384, "f16", dict(rms_eps=1e-6)),
                             ("f16 rms gelu", 1536, 384, "f16", dict(rms_eps=1e-6, epilogue=hip.EPI_GELU)), ("f16 rms glu", 768, 384, "f16", dict(rms_eps=1e-6, epilogue=hip.EPI_GLU)),
                             ("f16 argmax", 4096, 256, "f16", dict(c_mode=5)), ("f16 small", 96, 384, "f16", {}),
                             ("bf16 plain", 512, 512, "bf16", {}), ("bf16 res", 512, 2048, "bf16", dict(epilogue=hip.EPI_RES)), ("bf16 gelu", 2048, 512, "bf16", dict(epilogue=hip.EPI_GELU)),
                             ("x6 rms gelu", 512, 256, "x6", dict(rms_eps=1e-6, epilogue=hip.EPI_GELU))):
    for rows in (M, 200):
        A, W, b, Rr = rn(rows, K), rn(N, K, scale=K ** -0.5), rn(N, scale=0.1), rn(rows, N)
        Wp = hip.pack_w_f16x3(W) if kind == "f16" else (hip.pack_w_bf16x3(W) if kind == "bf16" else hip.pack_w_bf16x6(W))
        kw2 = dict(kw, M=rows, N=N, K=K, bias=b)
        if kw.get("epilogue") == hip.EPI_RES:
            kw2["R"] = Rr
        if kw.get("c_mode") == 5:
            C2 = torch.zeros(rows, (N + 63) // 64, 2, device=DEV)
            hip.gemm(A, Wp, torch.empty(1, device=DEV), C2=C2, ldc2=(N + 63) // 64, **kw2)
            out = C2
        else:
            out = torch.full((rows, N // 2 if kw.get("epilogue") == hip.EPI_GLU else N), float("nan"), device=DEV)
            hip.gemm(A, Wp, out, **kw2)
        torch.cuda.synchronize()
        print(f"{name:14s} {rows:5d} x {N} x {K}: {h(out)}", flush=True)
