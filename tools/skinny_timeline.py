#!/usr/bin/env python
"""Developer probe: in-kernel shader-clock stamps of the AR-step kernels (phases of one workgroup)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from sopro_amd import hip, pack

DEV = "cuda:0"
torch.manual_seed(0)
B, D = 32, 384


def run(name, nwg, **kw):
    dbg = torch.zeros(1024, 8, dtype=torch.int64, device=DEV)
    for _ in range(3):
        hip.skinny(dbg=dbg, **kw)
    torch.cuda.synchronize()
    d = dbg.cpu().double()
    d = d[d[:, 0] > 0]
    t0 = d[:, 0].min()
    rel = (d[:, :6] - t0)
    print(f"{name}: start spread {float(d[:,0].max()-t0):.0f} cyc | per-WG median phases (cycles since own start): "
          f"loads+stage {float((d[:,1]-d[:,0]).median()):.0f}, sync {float((d[:,2]-d[:,1]).median()):.0f}, mfma {float((d[:,3]-d[:,2]).median()):.0f}, "
          f"reduce {float((d[:,4]-d[:,3]).median()):.0f}, epilogue {float((d[:,5]-d[:,4]).median()):.0f} | last end {float(d[:,5].max()-t0):.0f} cyc")


X = torch.randn(B, D, device=DEV)
W1 = torch.randn(4 * D, D, device=DEV) * 0.05
W2 = torch.randn(D, 4 * D, device=DEV) * 0.05
Wg = torch.randn(2 * D, D, device=DEV) * 0.05
nw = torch.ones(D, device=DEV)
b1, b2, bg = torch.zeros(4 * D, device=DEV), torch.zeros(D, device=DEV), torch.zeros(2 * D, device=DEV)
U = torch.empty(B, 4 * D, device=DEV)
P = torch.empty(4, B, D, device=DEV)
Y = torch.empty(B, D, device=DEV)
ring = torch.zeros(25, B, D, device=DEV)
step = torch.zeros(1, dtype=torch.int32, device=DEV)
dww, dwb = torch.randn(13, D, device=DEV), torch.zeros(D, device=DEV)
run("FF1  (192 WG)", 192, X=X, W=W1, Y=U, B=B, N=4 * D, K=D, rms_norm=True, bias=b1, epilogue=hip.EPI_GELU)
run("FF2s (192 WG)", 192, X=U, W=W2, Y=P, B=B, N=D, K=4 * D, bias=b2, epilogue=hip.EPI_RES, R=X, ksplit=True, y_part_stride=B * D)
run("GLU  ( 48 WG)", 48, X=X, W=Wg, Y=Y, B=B, N=2 * D, K=D, rms_norm=True, bias=bg, epilogue=hip.EPI_GLU_DW, ring=ring, dw_w=dww, dw_b=dwb,
    step=step, ring_len=25, ring_bcap=B, dil=2, ksize=13)
run("GLUp ( 48 WG)", 48, X=P[0], W=Wg, Y=Y, B=B, N=2 * D, K=D, rms_norm=True, bias=bg, epilogue=hip.EPI_GLU_DW, ring=ring, dw_w=dww, dw_b=dwb,
    step=step, ring_len=25, ring_bcap=B, dil=2, ksize=13, Xp=P[1:], np_=3, xp_stride=B * D)
