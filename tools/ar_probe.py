#!/usr/bin/env python
"""Developer probe: time the recorded AR frame (hipGraph replay) in isolation.
    python tools/ar_probe.py [B] [steps] [greedy|sample]
Run under `rocprofv3 --kernel-trace --stats` for per-kernel durations."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

from bench import TEXT_LEN, build_engine, make_inputs
from sopro_amd.model import _ARRun

B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 200
mode = sys.argv[3] if len(sys.argv) > 3 else "sample"
tts, cfg, mc, wn, mn = build_engine("cuda:0")
ids, ref_tq = make_inputs(0)
ref = tts.prepare_reference(ref_tokens_tq=ref_tq)
prep = tts.model.prepare_conditioning_batch(ids[:B], [ref] * B, max_frames=steps - 1)
kw = dict(top_p=0.0, temperature=1.0, anti_loop=False) if mode == "greedy" else dict(top_p=0.9, temperature=1.05, anti_loop=True)
for rep in range(3):
    run = _ARRun(tts.model, prep["cond_ar"], prep["txt_seq"], prep["text_lens"], min_gen_frames=None, **kw)
    tts.model.stream.synchronize()
    t0 = time.perf_counter()
    run.advance(steps)
    tts.model.stream.synchronize()
    dt = time.perf_counter() - t0
    print(f"B={B} mode={mode} rep {rep}: {dt / steps * 1e6:.1f} us/frame, {run.plan.nlaunch} launches/frame")
