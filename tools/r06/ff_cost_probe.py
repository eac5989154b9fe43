#!/usr/bin/env python
"""r06: what the refinement's feed-forward contractions spend outside their MFMAs: the same 25600 x 1536 x 384 problem with / without the
GELU epilogue, with / without the fused RMSNorm staging, as f16 three-pass and as bf16 three-pass; ff2 (1536 -> 384, residual) and the
GLU projection likewise.  192-CU partition, HIP events over 40 launches.   python tools/r06/ff_cost_probe.py"""
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


def timed(fn, n=40):
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


M = 25600
for name, N, K in (("ff1", 1536, 384), ("glu", 768, 384), ("ff2", 384, 1536), ("mimi fc1", 2048, 512)):
    A, W, b, Rr = rn(M, K), rn(N, K, scale=K ** -0.5), rn(N, scale=0.1), rn(M, N)
    Cc = torch.empty(M, N, device=DEV)
    for kind, Wp in (("f16x3", hip.pack_w_f16x3(W)), ("bf16x3", hip.pack_w_bf16x3(W))):
        row = [f"{name:9s} {kind:7s} {M} x {N} x {K}:"]
        row.append(f"plain {timed(lambda: hip.gemm(A, Wp, Cc, M=M, N=N, K=K, bias=b)):7.1f}")
        row.append(f"gelu {timed(lambda: hip.gemm(A, Wp, Cc, M=M, N=N, K=K, bias=b, epilogue=hip.EPI_GELU)):7.1f}")
        row.append(f"res {timed(lambda: hip.gemm(A, Wp, Cc, M=M, N=N, K=K, bias=b, epilogue=hip.EPI_RES, R=Rr)):7.1f}")
        if kind == "f16x3" and K == 384:
            row.append(f"rms+plain {timed(lambda: hip.gemm(A, Wp, Cc, M=M, N=N, K=K, bias=b, rms_eps=1e-6)):7.1f}")
            row.append(f"rms+gelu {timed(lambda: hip.gemm(A, Wp, Cc, M=M, N=N, K=K, bias=b, rms_eps=1e-6, epilogue=hip.EPI_GELU)):7.1f}")
            if N % 64 == 0:
                Cg = torch.empty(M, N // 2, device=DEV)
                row.append(f"rms+glu {timed(lambda: hip.gemm(A, Wp, Cg, M=M, N=N, K=K, bias=b, rms_eps=1e-6, epilogue=hip.EPI_GLU)):7.1f}")
        print("  ".join(row) + "  us", flush=True)
