import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from sopro_amd import hip
DEV="cuda:0"
name,M,N,K=sys.argv[1],int(sys.argv[2]),int(sys.argv[3]),int(sys.argv[4])
g=torch.Generator(device=DEV).manual_seed(1)
A=torch.randn(M,K,device=DEV,generator=g); W=torch.randn(N,K,device=DEV,generator=g)*K**-0.5
Wp=hip.pack_w_bf16x3(W); C=torch.empty(M,N,device=DEV)
for _ in range(3): hip.gemm(A,Wp,C,M=M,N=N,K=K)
torch.cuda.synchronize()
