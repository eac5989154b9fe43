#!/usr/bin/env python
"""r06: the shader clock WHILE one kernel family runs back to back on the 192-CU partition (tools/micro/clock_probe.hip beside it):
which of the throughput partition's kernels are clock- (= power-) limited?  MFMA utilisation in CYCLES = MFMA cycles per launch /
(launch time x measured clock).   python tools/r06/kernel_clock_probe.py"""
import os
import sys
import time

R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, R)
sys.path.insert(0, os.path.join(R, "tools", "r06"))
import torch  # noqa: E402

from saturation_probe import Sampler, build_probe  # noqa: E402
from sopro_amd import hip  # noqa: E402

DEV = torch.device("cuda:0")
torch.cuda.set_device(0)
lib = build_probe()
sm = Sampler(lib, spin_us=100, period=0.001)
st = hip.cu_range_stream(64, 192, DEV)
g = torch.Generator(device=DEV).manual_seed(1)


def rn(*shape, scale=1.0):
    return torch.randn(*shape, device=DEV, generator=g) * scale


def loop(name, fn, mfma_cycles_per_cu, seconds=0.6):
    """mfma_cycles_per_cu: MFMA pipe cycles per SIMD one launch needs on 192 CUs (passes x flops / (192 CUs x 4 SIMDs x 1024 flop/cycle))"""
    with torch.cuda.stream(st):
        for _ in range(2):
            fn()
        torch.cuda.synchronize()
        sm.start()
        n, t0 = 0, time.perf_counter()
        while time.perf_counter() - t0 < seconds:
            fn()
            n += 1
            if n % 8 == 0:
                st.synchronize()
        st.synchronize()
        dt = (time.perf_counter() - t0) / n
    rows = sm.rows[:]
    s = sm.stop()
    import numpy as np

    mhz = float(np.median([r[1] for r in rows])) if rows else 0.0
    util = mfma_cycles_per_cu / (dt * mhz * 1e6) if mhz else 0.0
    print(f"{name:28s} {dt * 1e6:9.1f} us per launch   clock p50 {mhz:5.0f} MHz   MFMA pipe busy (in cycles) {util:5.2f}   | {s[:60]}", flush=True)


def mfma_cyc(flops_fp32eq, passes=3):
    return passes * flops_fp32eq / (192 * 4 * 1024.0)


# ---- fused last level (uptail) and the 128-channel residual block at the pipeline's pass shape
B, T = 64, 96000
x = torch.nn.functional.elu(rn(B, 1 + T, 128)); x[:, 0] = 0
W, bu = rn(256, 256, scale=0.06), rn(256, scale=0.1)
w1, b1, w2, b2, wf = rn(32, 192, scale=0.07), rn(32, scale=0.1), rn(64, 32, scale=0.17), rn(64, scale=0.1), rn(3, 64, scale=0.07)
wav = torch.empty(B, 4 * T, device=DEV)
fl = 2.0 * B * T * 256 * 256 + 2.0 * B * 4 * T * (3 * 64 * 32 + 32 * 64 + 3 * 64)
loop("seanet_uptail (64 x 96000)", lambda: hip.seanet_uptail(x, W, bu, w1, b1, w2, b2, wf, 0.03, wav, B=B, T=T, x_seg_stride=(1 + T) * 128, wav_seg_stride=4 * T), mfma_cyc(fl))
del wav
hh = torch.zeros(B, 2 + T, 128, device=DEV); hh[:, 2:] = rn(B, T, 128)
ho = torch.zeros(B, 2 + T, 128, device=DEV)
r1, rb1, r2, rb2 = rn(64, 384, scale=0.05), rn(64, scale=0.1), rn(128, 64, scale=0.12), rn(128, scale=0.1)
loop("seanet_res128 (64 x 96000)", lambda: hip.seanet_res128(hh, r1, rb1, r2, rb2, ho, B=B, T=T, h_seg_stride=(2 + T) * 128, out_seg_stride=(2 + T) * 128),
     mfma_cyc(2.0 * B * T * (3 * 128 * 64 + 64 * 128)))
del hh, ho, x
# ---- contractions: tile kernel and long-K form
for name, M, N, K, long_k in (("up2 tile 1228800x640x512", 1228800, 640, 512, False), ("up1 tile 204800x1536x1024", 204800, 1536, 1024, False),
                              ("up1 long-K", 204800, 1536, 1024, True), ("fc1 tile 25600x2048x512", 25600, 2048, 512, False),
                              ("fc2 tile 25600x512x2048", 25600, 512, 2048, False)):
    A = torch.nn.functional.elu(rn(M, K))
    Wm = rn(N, K, scale=K ** -0.5)
    Wp = hip.pack_w_bf16x3(Wm, rows=long_k)
    Cc = torch.empty(M, N, device=DEV)
    if long_k:
        gsz = A.reshape(M, K // 32, 32)
        hi = gsz.to(torch.bfloat16)
        lo = (gsz - hi.float()).to(torch.bfloat16)
        A = torch.cat([hi, lo], dim=-1).contiguous().view(torch.float32).reshape(M, K)
        loop(name, lambda: hip.gemm(A, Wp, Cc, M=M, N=N, K=K, a_split=True, long_k=True), mfma_cyc(2.0 * M * N * K))
    else:
        loop(name, lambda: hip.gemm(A, Wp, Cc, M=M, N=N, K=K), mfma_cyc(2.0 * M * N * K))
    del A, Wm, Wp, Cc
# ---- refinement feed-forward (f16 three-pass, fused RMSNorm + GELU)
M, N, K = 25600, 1536, 384
A, Wm = rn(M, K), rn(N, K, scale=K ** -0.5)
Wp = hip.pack_w_f16x3(Wm)
Cc = torch.empty(M, N, device=DEV)
loop("nar ff1 f16x3 25600x1536x384", lambda: hip.gemm(A, Wp, Cc, M=M, N=N, K=K, epilogue=hip.EPI_GELU, rms_eps=1e-6), mfma_cyc(2.0 * M * N * K))
