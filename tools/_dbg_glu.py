import sys; sys.path.insert(0,'/root/repo')
import torch
from sopro_amd import hip
DEV='cuda:0'; B,D,k,dil=5,384,13,2; L=(k-1)*dil+1
torch.manual_seed(0)
x=torch.randn(B,D,device=DEV); W=torch.randn(2*D,D,device=DEV)*0.05; b=torch.zeros(2*D,device=DEV)
ring=torch.zeros(L,B,D,device=DEV); step=torch.full((1,),3,dtype=torch.int32,device=DEV)
dww=torch.ones(13,D,device=DEV); dwb=torch.zeros(D,device=DEV)
Y=torch.full((B,D),-7.0,device=DEV)
hip.skinny(x,W,Y,B=B,N=2*D,K=D,rms_norm=True,eps=1e-6,bias=b,epilogue=hip.EPI_GLU_DW,ring=ring,dw_w=dww,dw_b=dwb,step=step,ring_len=L,ring_bcap=B,dil=dil,ksize=k)
torch.cuda.synchronize()
print('nan count Y', int(torch.isnan(Y).sum()), 'of', Y.numel())
print('nan rows', torch.isnan(Y).any(1).tolist(), 'nan cols(first 20)', torch.isnan(Y).any(0)[:20].tolist())
print('ring nonzero slots', [int(s) for s in (ring.abs().sum((1,2))>0).nonzero().flatten()], 'ring nan', int(torch.isnan(ring).sum()))
xn=x*torch.rsqrt((x*x).mean(1,keepdim=True)+1e-6); g=xn@W.t(); h=g[:,:D]*torch.sigmoid(g[:,D:])
print('h err vs ring slot 3', float((ring[3]-h).abs().max()))
print('Y ref err', float((Y-(x+h)).abs().max()))
