#!/usr/bin/env python
"""Developer probe: where the time-to-first-audio of stream() goes (host wall time with a device sync after each stage)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import bench
from sopro_amd.codec import MimiDecodeState

tts, cfg, mc, wn, mn = bench.build_engine("cuda:0")
ids, ref_tq = bench.make_inputs(0)
ref = tts.prepare_reference(ref_tokens_tq=ref_tq)
m = tts.model
acc = {}
N = 30
for it in range(N + 5):
    t = [time.perf_counter()]
    def mark():
        torch.cuda.synchronize(); t.append(time.perf_counter())
    prep = m.prepare_conditioning(ids[it % 32], ref, max_frames=199, style_strength=float(cfg.style_strength)); mark()
    gen = m.ar_stream(prep, max_frames=199, top_p=0.9, temperature=1.05, anti_loop=True, lookahead=6)
    hist = []
    for _t, tok, eos in gen:
        hist.append(int(tok))
        if len(hist) == 6:
            break
    mark()
    toks = m.nar_refine(prep["cond_ar"][:, 0:6, :], torch.as_tensor(hist, dtype=torch.long).unsqueeze(0)).squeeze(0); mark()
    from sopro_amd.codec import MimiStreamDecoder
    wav, st = MimiStreamDecoder(tts.codec).decode_step(toks, MimiDecodeState()); mark()
    del gen
    if it >= 5:
        for k, (a, b) in zip(("cond", "ar6", "nar", "mimi"), zip(t[:-1], t[1:])):
            acc.setdefault(k, []).append((b - a) * 1e3)
print({k: round(float(np.median(v)), 3) for k, v in acc.items()}, "sum", round(sum(float(np.median(v)) for v in acc.values()), 3))
