#!/usr/bin/env python
"""Developer probe (timing build tools/micro/gemm_kstep_stamps.patch -> SOPRO_HIP_LIB=tools/micro/libsopro_gemm_stamps.so): where a wave
of the three-pass / one-pass contraction spends a K-step.  Wave 0 of every workgroup sums shader clocks over its K loop:
  issue  = issuing the next steps' global requests (A rows two steps ahead, W fragments one)
  lds    = from issuing a substep's fragment reads to their arrival (s_waitcnt lgkmcnt(0))
  mfma   = from there to the last MFMA of the substep being ISSUED (a busy pipe stalls the issue)
  stage  = splitting / storing the next step's A rows into LDS (includes waiting for their global loads)
  bar    = the workgroup barrier
Ticks are not a time base (profiles/r04_experiments.md) - the split in per cent is the point."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from sopro_amd import hip

DEV = "cuda:0"
shapes = [("up0", 12800, 4096, 2048), ("tr.qkv", 12800, 1536, 512), ("tr.fc2", 12800, 512, 2048), ("up1", 102400, 1536, 1024), ("up2", 614400, 640, 512)]
lib = hip.load()
if os.environ.get("PROBE_CUS"):
    n_cus = int(os.environ["PROBE_CUS"])
    torch.cuda.set_stream(hip.cu_range_stream(256 - n_cus, n_cus, torch.device(DEV)))
for name, M, N, K in shapes:
    g = torch.Generator(device=DEV).manual_seed(1)
    A = torch.randn(M, K, device=DEV, generator=g)
    W = torch.randn(N, K, device=DEV, generator=g) * K ** -0.5
    C = torch.empty(M, N, device=DEV)
    for pieces, Wp in ((2, hip.pack_w_bf16x3(W)), (1, hip.pack_w_bf16x1(W))):
        lib.sopro_gemm_bf16_set_tile_override(1 if pieces == 2 else 0)
        ntile = ((M + 127) // 128) * ((N + 127) // 128)
        dbg = torch.zeros(ntile, 8, dtype=torch.int64, device=DEV)
        for _ in range(2):
            hip.gemm(A, Wp, C, M=M, N=N, K=K, dbg=dbg)
        torch.cuda.synchronize()
        d = dbg.double().cpu()
        loop = (d[:, 2] - d[:, 1]).mean()
        parts = {k: float(d[:, i].mean() / loop * 100) for k, i in (("issue", 3), ("lds", 4), ("mfma", 5), ("stage", 6), ("bar", 7))}
        tot = float((d[:, 2] - d[:, 0]).mean())
        print(f"{name:7s} {'bf16x3' if pieces == 2 else 'bf16x1'} K loop = {loop / tot * 100:4.1f} % of the workgroup's ticks up to the epilogue; of the loop: "
              + "  ".join(f"{k} {v:4.1f} %" for k, v in parts.items()) + f"  (sum {sum(parts.values()):5.1f} %; {loop / ((K + 31) // 32):6.0f} ticks per K-step)", flush=True)
    lib.sopro_gemm_bf16_set_tile_override(0)
