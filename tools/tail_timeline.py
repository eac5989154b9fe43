#!/usr/bin/env python
"""Developer probe (round 3): where a tile of the fused SEANet tail kernel (sixteen-wave form) spends its time - shader-clock stamps of one tile group
(timeline build: `bash tools/micro/build_tail_dbg.sh`, then SOPRO_HIP_LIB=tools/micro/libsopro_taildbg.so python tools/tail_timeline.py).
Stamps per tile: 0 loop top (tile requested next), 1 tile arrived, 2 staged (ELU + split written), 3 barrier passed, 4 first
convolution done, 5 barrier, 6 second convolution + ELU(h') written, 7 barrier; the last convolution runs from 7 to the next 0."""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from sopro_amd import hip

B, T = int(os.environ.get("PROBE_B", 32)), 384000
dev = "cuda:0"
g = torch.Generator(device=dev).manual_seed(0)
h = torch.randn(B, 2 + T, 64, device=dev, generator=g)
h[:, :2] = 0
w1, b1 = torch.randn(32, 192, device=dev, generator=g) * 0.07, torch.randn(32, device=dev, generator=g) * 0.1
w2, b2 = torch.randn(64, 32, device=dev, generator=g) * 0.17, torch.randn(64, device=dev, generator=g) * 0.1
wf = torch.randn(3, 64, device=dev, generator=g) * 0.07
wav = torch.empty(B, T, device=dev)
dbg = torch.zeros(16 * 8, dtype=torch.int64, device=dev)
lib = hip.load()
lib.sopro_tail_dbg_set.argtypes = [C.c_void_p]
assert lib.sopro_tail_dbg_set(dbg.data_ptr()) == 0
for _ in range(3):
    hip.seanet_tail(h, w1, b1, w2, b2, wf, 0.03, wav, B=B, T=T, h_seg_stride=(2 + T) * 64, wav_seg_stride=T)
torch.cuda.synchronize()
d = dbg.cpu().view(16, 8)
names = ["tile wait", "ELU + split", "barrier", "conv1 + epilogue", "barrier", "conv2 + epilogue", "barrier", "last conv + loop"]
tot = [0.0] * 8
n = 0
last = max(i for i in range(16) if int(d[i][7]) != 0)  # trips the stamped workgroup ran
for it in range(1, last):
    st = [int(v) for v in d[it]] + [int(d[it + 1][0])]
    for i in range(8):
        tot[i] += st[i + 1] - st[i]
    n += 1
print(f"B={B}: cycles per tile of one workgroup (mean of {n} tiles): total {sum(tot) / n:.0f}")
for nm, v in zip(names, tot):
    print(f"   {nm:18s} {v / n:8.0f}")
