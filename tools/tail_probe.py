#!/usr/bin/env python
"""Developer probe: the fused SEANet tail kernel alone at the bench shape (32 utterances x 200 frames = 384000 samples each).
Run under `rocprofv3 --pmc ... --kernel-trace` for its SQ counters."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from sopro_amd import hip

B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
T = int(sys.argv[2]) if len(sys.argv) > 2 else 384000
dev = "cuda:0"
if len(sys.argv) > 3:  # tiles per workgroup of the four-wave kernel (0 = the library's choice; -1 = the sixteen-wave kernel, -N = with N trips per workgroup)
    hip.load().sopro_seanet_tail_set_tiles(int(sys.argv[3]))
g = torch.Generator(device=dev).manual_seed(0)
h = torch.randn(B, 2 + T, 64, device=dev, generator=g)
h[:, :2] = 0
w1 = torch.randn(32, 192, device=dev, generator=g) * 0.07
b1 = torch.randn(32, device=dev, generator=g) * 0.1
w2 = torch.randn(64, 32, device=dev, generator=g) * 0.17
b2 = torch.randn(64, device=dev, generator=g) * 0.1
wf = torch.randn(3, 64, device=dev, generator=g) * 0.07
wav = torch.empty(B, T, device=dev)
for _ in range(2):
    hip.seanet_tail(h, w1, b1, w2, b2, wf, 0.03, wav, B=B, T=T, h_seg_stride=(2 + T) * 64, wav_seg_stride=T)
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(5):
    hip.seanet_tail(h, w1, b1, w2, b2, wf, 0.03, wav, B=B, T=T, h_seg_stride=(2 + T) * 64, wav_seg_stride=T)
e1.record()
torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / 5
print(f"B={B} T={T}: {ms * 1e3:.1f} us  ({B * T * 64 * 4 / ms / 1e6:.0f} GB/s of h, {B * T * 16832 / ms / 1e9:.1f} TFLOP/s fp32-equivalent)")
