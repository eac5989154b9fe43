#!/usr/bin/env python
"""Developer probe (round 3): the codec decoder alone on the whole chip, per kernel family (the library's own launch timing,
csrc/prof.hip): ms per family and launch for a batch of PROBE_B x 200 frames.
    PROBE_B=32 python tools/mimi_probe.py"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from bench import build_engine
from sopro_amd import hip

tts, cfg, mc, wn, mn = build_engine("cuda:0")
for B in [int(v) for v in os.environ.get("PROBE_B", "32,64").split(",")]:
    toks = torch.randint(0, 2048, (B, 200, 32), device=tts.device)
    for _ in range(2):
        tts.codec.decode_batch(toks)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    with torch.cuda.stream(tts.codec.stream):
        e0.record()
        for _ in range(3):
            tts.codec.decode_batch(toks)
        e1.record()
    torch.cuda.synchronize()
    print(f"B={B}: decode (recorded sequence) {e0.elapsed_time(e1) / 3:.3f} ms", flush=True)
    p = hip.Profiler()
    hip.set_profiler(p)
    for _ in range(3):
        tts.codec.decode_batch(toks)
    torch.cuda.synchronize()
    hip.set_profiler(None)
    fam = p.summary()
    for k, v in sorted(fam.items(), key=lambda kv: -kv[1]["ms_all"]):
        print(f"   {k:24s} {v['launches'] // 3:4d} launches  {v['ms_all'] / 3:8.3f} ms  {v['flops'] / max(v['ms_all'], 1e-9) / 1e9:8.1f} TFLOP/s", flush=True)
