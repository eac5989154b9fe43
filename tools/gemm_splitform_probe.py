#!/usr/bin/env python
"""Developer probe: the three-pass GEMM with fp32 A rows (split on the fly while staged) against A rows already in split
form (pure copies into LDS), same shapes, same output mode."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from sopro_amd import hip
DEV="cuda:0"
shapes=[("up0",12800,4096,2048),("conv0",12800,1024,3584),("res0.c1",102400,256,1536),("up1",102400,1536,1024),("up2",614400,640,512),("up3",3072000,256,256)]
for name,M,N,K in shapes:
    g=torch.Generator(device=DEV).manual_seed(1)
    A=torch.randn(M,K,device=DEV,generator=g); W=torch.randn(N,K,device=DEV,generator=g)*K**-0.5; b=torch.randn(N,device=DEV,generator=g)
    # a valid split-form image: bf16 hi/lo pairs of small numbers
    Ab=(torch.randn(M,K*2,device=DEV,generator=g)*0.1).to(torch.bfloat16).view(torch.float32).contiguous()  # [M, K] fp32-sized container
    Wp=hip.pack_w_bf16x3(W); C=torch.empty(M,N,device=DEV)
    res=[]
    for label,kw,src in (("fp32A",dict(),A),("splitA",dict(a_split=True),Ab)):
        for _ in range(2): hip.gemm(src,Wp,C,M=M,N=N,K=K,bias=b,**kw)
        e0,e1=torch.cuda.Event(enable_timing=True),torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(5): hip.gemm(src,Wp,C,M=M,N=N,K=K,bias=b,**kw)
        e1.record(); torch.cuda.synchronize()
        us=e0.elapsed_time(e1)/5*1e3
        res.append(f"{label}: {us:8.1f}us {2.0*M*N*K/us/1e6:6.1f}TF")
    print(f"{name:8s} M={M:8d} N={N:5d} K={K:5d} | "+" | ".join(res),flush=True)
