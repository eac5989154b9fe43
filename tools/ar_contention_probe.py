#!/usr/bin/env python
"""Developer probe: WHAT on the throughput partition slows the AR frame down?
Two AR phases replay their frame graphs on the shared 64-CU partition while a background thread keeps the other 192 CUs busy with
one kind of synthetic load:
  none     - nothing
  stream   - a copy of a 1 GiB buffer over and over (pure HBM streaming, evicts the Infinity Cache)
  l2copy   - a copy of a 2 MiB buffer over and over (many short kernels whose data stays in L2: dispatch + CU activity only)
  mfma     - the three-pass contraction on a 1024 x 1024 x 1024 problem (12 MB of operands: cache resident, matrix cores busy)
  gemmbig  - the three-pass contraction on the decoder's up1 shape (102400 x 1536 x 1024: matrix cores + 1 GB of traffic per launch)
Prints us per AR frame of each phase."""
import os
import sys
import threading
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from bench import build_engine, make_inputs
from sopro_amd import hip
from sopro_amd.model import _ARRun

B, steps = 32, 200
tts, cfg, mc, wn, mn = build_engine("cuda:0")
dev = tts.device
ids, ref_tq = make_inputs(0)
ref = tts.prepare_reference(ref_tokens_tq=ref_tq)
lanes = [tts, tts.clone_lane()]
kw = dict(top_p=0.9, temperature=1.05, anti_loop=True)
preps = [l.model.prepare_conditioning_batch(ids[:B], [ref] * B, max_frames=steps - 1) for l in lanes]
if os.environ.get("PROBE_ALIAS") == "1":  # what would L2-hot weights buy? every block reads block 0's matrices (0.7 MB per XCD)
    for l in lanes:
        for i in range(1, 6):
            for suf in (".glu.w", ".ff1.w", ".ff2.w"):
                l.model.wk[f"ar.blocks.{i}{suf}"] = l.model.wk[f"ar.blocks.0{suf}"]
for l in lanes:
    s = hip.cu_range_stream(0, 64, dev)
    l.model.stream = l.model.prep_stream = l.model.bulk_stream = s
    l.model._ar_cache.clear()
bulk = hip.cu_range_stream(64, 192, dev)
big_a, big_b = torch.empty(1 << 28, device=dev), torch.empty(1 << 28, device=dev)
sm_a, sm_b = torch.empty(1 << 19, device=dev), torch.empty(1 << 19, device=dev)
A1, W1, C1 = torch.randn(1024, 1024, device=dev), hip.pack_w_bf16x3(torch.randn(1024, 1024, device=dev) * 0.03), torch.empty(1024, 1024, device=dev)
A2, W2, C2 = torch.randn(102400, 1024, device=dev), hip.pack_w_bf16x3(torch.randn(1536, 1024, device=dev) * 0.03), torch.empty(102400, 1536, device=dev)
torch.cuda.synchronize()
stop = threading.Event()


def background(kind):
    with torch.cuda.stream(bulk):
        n = 0
        while not stop.is_set():
            for _ in range(8):
                if kind == "stream":
                    big_b.copy_(big_a)
                elif kind == "l2copy":
                    for _ in range(16):
                        sm_b.copy_(sm_a)
                elif kind == "mfma":
                    for _ in range(16):
                        hip.gemm(A1, W1, C1, M=1024, N=1024, K=1024)
                elif kind == "gemmbig":
                    hip.gemm(A2, W2, C2, M=102400, N=1536, K=1024)
            n += 1
            if n % 4 == 0:
                bulk.synchronize()  # stay a bounded distance ahead of the GPU
        bulk.synchronize()


EMPTY = len(sys.argv) > 1 and sys.argv[1] == "empty"  # chains of 23 one-workgroup kernels instead of the frame: the launch path alone
tiny = [torch.zeros(64, device=dev) for _ in range(2)]
empty_graphs = []
if EMPTY:
    for l in lanes:
        with torch.cuda.stream(l.model.stream):
            hip.tanh_affine(tiny[0], tiny[1], 0.0, 1.0, 64)
            l.model.stream.synchronize()
            hip.capture_begin()
            for _ in range(23):
                hip.tanh_affine(tiny[0], tiny[1], 0.0, 1.0, 64)
            empty_graphs.append(hip.capture_end())


def phase(lane, prep, out, i, bar):
    with torch.cuda.stream(lane.model.stream):
        run = None if EMPTY else _ARRun(lane.model, prep["cond_ar"], prep["txt_seq"], prep["text_lens"], min_gen_frames=None, **kw)
        lane.model.stream.synchronize()
        bar.wait()
        t0 = time.perf_counter()
        if EMPTY:
            for _ in range(steps):
                empty_graphs[i].launch()
        else:
            run.advance(steps)
        lane.model.stream.synchronize()
        out[i] = (time.perf_counter() - t0) / steps * 1e6


for kind in ("none", "stream", "l2copy", "mfma", "gemmbig", "none"):
    stop.clear()
    bg = None
    if kind != "none":
        bg = threading.Thread(target=background, args=(kind,))
        bg.start()
        time.sleep(0.3)
    res = []
    for rep in range(3):
        out = [0.0, 0.0]
        bar = threading.Barrier(2)
        th = [threading.Thread(target=phase, args=(lanes[i], preps[i], out, i, bar)) for i in range(2)]
        for t in th:
            t.start()
        for t in th:
            t.join()
        res.append("/".join(f"{o:6.1f}" for o in out))
    stop.set()
    if bg is not None:
        bg.join()
    print(f"background on the other 192 CUs: {kind:8s} us/frame per phase (2 phases on 64 CUs): " + "   ".join(res), flush=True)
