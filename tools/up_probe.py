#!/usr/bin/env python
"""Developer probe: the weight-stationary last transposed convolution of SEANet (sopro_seanet_up128_f32) at the bench shape
(32 utterances x 96000 input rows) against the generic tile kernel, for several tiles-per-workgroup settings."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from sopro_amd import hip

DEV = "cuda:0"
B, T, ci, co, r = 32, int(sys.argv[1]) if len(sys.argv) > 1 else 96000, 128, 64, 4
g = torch.Generator(device=DEV).manual_seed(1)
x = torch.randn(B, 1 + T, ci, device=DEV, generator=g)
x[:, 0] = 0
W = torch.randn(r * co, 2 * ci, device=DEV, generator=g) * 0.06
b = torch.randn(r * co, device=DEV, generator=g)
out = torch.empty(B, 2 + T * r, co, device=DEV)
lib = hip.load()


def timed(fn, n=5):
    for _ in range(2):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


flop = 2.0 * B * T * 256 * 256
for passes, Wp in ((3, hip.pack_w_bf16x3(W)), (1, hip.pack_w_bf16x1(W))):
    us = timed(lambda: hip.gemm(x, Wp, out, M=B * T, N=r * co, K=2 * ci, lda=ci, bias=b, rows_per_seg=T, a_seg_stride=(1 + T) * ci, c_off=2 * co,
                                c_seg_stride=(2 + T * r) * co, ldc=r * co))
    print(f"passes {passes}: tile kernel {us:8.1f} us {flop / us / 1e6:6.1f} TF", flush=True)
    for tiles in (0, 1, 4, 12, 24, 47, 94, 188):
        lib.sopro_seanet_up_set_tiles(tiles)
        us = timed(lambda: hip.seanet_up128(x, W, b, out, B=B, T=T, x_seg_stride=(1 + T) * ci, out_seg_stride=(2 + T * r) * co, out_off=2 * co, passes=passes))
        print(f"passes {passes}: weight-stationary, tiles {tiles:3d}: {us:8.1f} us {flop / us / 1e6:6.1f} TF  ({(B * (1 + T) * ci + B * T * 256) * 4 / us / 1e6:5.2f} TB/s)", flush=True)
    lib.sopro_seanet_up_set_tiles(0)
