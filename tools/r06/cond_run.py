#!/usr/bin/env python
"""r06: the conditioning phase alone (128 utterances, S = 64, Tr = 150, 200 frames) on a CU-masked stream - for a rocprofv3 kernel trace.
    python tools/r06/cond_run.py [cus] [reps]"""
import os
import sys
import time

R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, R)
import torch  # noqa: E402

import bench  # noqa: E402
from sopro_amd import hip  # noqa: E402

cus = int(sys.argv[1]) if len(sys.argv) > 1 else 192
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 4
torch.cuda.set_device(0)
tts, cfg, mc, wn, mn = bench.build_engine("cuda:0", "f32")
m = tts.model
ids, _ = bench.make_inputs(0)
voices = [tts.prepare_reference(ref_tokens_tq=v) for v in bench.make_voices(0, 32)]
idsB, refsB = ids * 4, voices * 4
if cus < 256:
    m.prep_stream = hip.cu_range_stream(256 - cus, cus, torch.device("cuda:0"))
for _ in range(3):
    prep = m.phase_cond(idsB, refsB, max_frames=199, style_strength=float(cfg.style_strength))
    run = m.ar_prepare(prep, top_p=0.9, temperature=1.05, anti_loop=True, min_gen_frames=None, seed=1)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(reps):
    prep = m.phase_cond(idsB, refsB, max_frames=199, style_strength=float(cfg.style_strength))
    run = m.ar_prepare(prep, top_p=0.9, temperature=1.05, anti_loop=True, min_gen_frames=None, seed=1)
torch.cuda.synchronize()
print(f"conditioning + AR preparation of 128 utterances on {cus} CUs: {(time.perf_counter() - t0) / reps * 1e3:.3f} ms")
