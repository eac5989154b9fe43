#!/usr/bin/env python
"""Developer probe: AR frame time when the engine stream is confined to a CU range, alone and next to a GEMM load."""
import os, sys, time, threading
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from bench import build_engine, make_inputs
from sopro_amd import hip
from sopro_amd.model import _ARRun

tts, cfg, mc, wn, mn = build_engine("cuda:0")
ids, ref_tq = make_inputs(0)
ref = tts.prepare_reference(ref_tokens_tq=ref_tq)
B, steps = 32, 200
prep = tts.model.prepare_conditioning_batch(ids[:B], [ref] * B, max_frames=steps - 1)
kw = dict(top_p=0.9, temperature=1.05, anti_loop=True)
A = torch.randn(200000, 512, device="cuda:0"); W = torch.randn(512, 512, device="cuda:0") * 0.04; Cc = torch.empty(200000, 512, device="cuda:0")

def ar_time(stream, with_load=None):
    m = tts.model
    old = m.stream
    m.stream = stream
    m._ar_cache.clear()
    try:
        run = _ARRun(m, prep["cond_ar"], prep["txt_seq"], prep["text_lens"], min_gen_frames=None, **kw)
        run.advance(20); stream.synchronize()
        stop = threading.Event()
        th = None
        if with_load is not None:
            def bg():
                with torch.cuda.stream(with_load):
                    while not stop.is_set():
                        for _ in range(20):
                            hip.gemm(A, W, Cc, M=200000, N=512, K=512)
                        with_load.synchronize()
            th = threading.Thread(target=bg); th.start(); time.sleep(0.05)
        t0 = time.perf_counter(); run.advance(steps - 20); stream.synchronize(); dt = time.perf_counter() - t0
        if th: stop.set(); th.join()
        return dt / (steps - 20) * 1e6
    finally:
        m.stream = old
        m._ar_cache.clear()

full = torch.cuda.Stream()
print("full chip, alone        : %.1f us/frame" % ar_time(full))
for n in (32, 64, 96):
    s = hip.cu_range_stream(0, n)
    print(f"AR on CUs [0,{n}), alone  : %.1f us/frame" % ar_time(s))
    rest = hip.cu_range_stream(n, 256 - n)
    print(f"AR on CUs [0,{n}) + GEMM on the other {256-n}: %.1f us/frame" % ar_time(s, rest))
print("AR full chip + GEMM full chip (no masks): %.1f us/frame" % ar_time(full, torch.cuda.Stream()))
