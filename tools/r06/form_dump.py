import os, sys
sys.path.insert(0, "/root/repo")
import torch
from sopro_amd import hip
DEV = torch.device("cuda:0"); torch.cuda.set_device(0)
g = torch.Generator(device=DEV).manual_seed(7)
rn = lambda *s, scale=1.0: torch.randn(*s, device=DEV, generator=g) * scale
M, N, K = 4136, 768, 384
A, W, b = rn(M, K), rn(N, K, scale=K ** -0.5), rn(N, scale=0.1)
Wp = hip.pack_w_f16x3(W)
out = torch.full((M, N // 2), float("nan"), device=DEV)
hip.gemm(A, Wp, out, M=M, N=N, K=K, bias=b, rms_eps=1e-6, epilogue=hip.EPI_GLU)
pre = torch.full((M, N), float("nan"), device=DEV)
hip.gemm(A, Wp, pre, M=M, N=N, K=K, bias=b, rms_eps=1e-6)
torch.cuda.synchronize()
torch.save({"glu": out.cpu(), "pre": pre.cpu()}, sys.argv[1])
