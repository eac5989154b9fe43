#!/usr/bin/env python
"""Developer probe: the last SEANet level as one kernel (sopro_seanet_uptail_*) against the two kernels it replaces
(sopro_seanet_up128_* + sopro_seanet_tail_*) at the pipeline's pass shape (64 utterances x 96000 input rows by default), on the
whole chip and on the 192-CU throughput partition, three-pass fp32 rows and one-pass bf16 rows, for several tiles-per-workgroup
settings.  usage: uptail_probe.py [B] [T] [fused]   (fused: only the one-kernel form at its default tiling - the ablation builds of
the ablation builds of round 4 - removed in round 5, last tree 9bb62d2; their numbers: profiles/r04_uptail_ablation.txt)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from sopro_amd import hip

DEV = torch.device("cuda:0")
B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
T = int(sys.argv[2]) if len(sys.argv) > 2 else 96000
FUSED_ONLY = len(sys.argv) > 3 and sys.argv[3] == "fused"
ci, co, r = 128, 64, 4
S = r * T
g = torch.Generator(device=DEV).manual_seed(1)
x = torch.nn.functional.elu(torch.randn(B, 1 + T, ci, device=DEV, generator=g))
x[:, 0] = 0
x16 = x.to(torch.bfloat16)
W = torch.randn(r * co, 2 * ci, device=DEV, generator=g) * 0.06
bu = torch.randn(r * co, device=DEV, generator=g) * 0.1
w1, b1 = torch.randn(32, 192, device=DEV, generator=g) * 0.07, torch.randn(32, device=DEV, generator=g) * 0.1
w2, b2 = torch.randn(64, 32, device=DEV, generator=g) * 0.17, torch.randn(64, device=DEV, generator=g) * 0.1
wf = torch.randn(3, 64, device=DEV, generator=g) * 0.07
h = torch.zeros(B, 2 + S, co, device=DEV)
h16 = torch.zeros(B, 2 + S, co, device=DEV, dtype=torch.bfloat16)
wav_a = torch.empty(B, S, device=DEV)
wav_b = torch.empty(B, S, device=DEV)
lib = hip.load()
flop = 2.0 * B * T * 256 * 256 + 2.0 * B * S * (3 * 64 * 32 + 32 * 64 + 3 * 64)


def timed(fn, stream, n=4):
    with torch.cuda.stream(stream):
        for _ in range(2):
            fn()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(stream)
        for _ in range(n):
            fn()
        e1.record(stream)
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


def two32():
    hip.seanet_up128(x, W, bu, h, B=B, T=T, x_seg_stride=(1 + T) * ci, out_seg_stride=(2 + S) * co, out_off=2 * co, passes=3)
    hip.seanet_tail(h, w1, b1, w2, b2, wf, 0.03, wav_a, B=B, T=S, h_seg_stride=(2 + S) * co, wav_seg_stride=S)


def two16():
    hip.seanet_up128_bf16(x16, W, bu, h16, B=B, T=T, x_seg_stride=(1 + T) * ci, out_seg_stride=(2 + S) * co, out_off=2 * co)
    hip.seanet_tail_bf16(h16, w1, b1, w2, b2, wf, 0.03, wav_a, B=B, T=S, h_seg_stride=(2 + S) * co, wav_seg_stride=S)


def one32():
    hip.seanet_uptail(x, W, bu, w1, b1, w2, b2, wf, 0.03, wav_b, B=B, T=T, x_seg_stride=(1 + T) * ci, wav_seg_stride=S)


def one16():
    hip.seanet_uptail(x16, W, bu, w1, b1, w2, b2, wf, 0.03, wav_b, B=B, T=T, x_seg_stride=(1 + T) * ci, wav_seg_stride=S)


total = hip.device_info(0)["cus"]
streams = [("whole chip", torch.cuda.Stream(device=DEV)), ("192-CU partition", hip.cu_range_stream(64, total - 64, DEV))]
for name, st in streams:
    for what, two, one in (("three-pass, fp32 rows", two32, one32), ("one pass, bf16 rows", two16, one16)):
        if FUSED_ONLY:
            print(f"{name:18s} {what:22s} one kernel ({os.path.basename(os.environ.get('SOPRO_HIP_LIB', 'product'))}): {timed(one, st):9.1f} us", flush=True)
            continue
        us2 = timed(two, st)
        print(f"{name:18s} {what:22s} two kernels {us2:9.1f} us  {flop / us2 / 1e6:6.1f} TF", flush=True)
        for tiles in (0, 24, 47, 94, 375):
            lib.sopro_seanet_uptail_set_tiles(tiles)
            us1 = timed(one, st)
            print(f"{name:18s} {what:22s} one kernel, tiles {tiles:3d}: {us1:9.1f} us  {flop / us1 / 1e6:6.1f} TF  ({us2 / us1:4.2f}x)", flush=True)
        lib.sopro_seanet_uptail_set_tiles(0)
        d = float((wav_a - wav_b).abs().max()), float(wav_a.abs().max())
        print(f"{name:18s} {what:22s} max |difference| {d[0]:.3e} of peak {d[1]:.3e}", flush=True)
hip.destroy_stream(streams[1][1])
