#!/usr/bin/env python
"""Developer probe: a ragged workload (frame budgets 80..400) through static batches of 32 (every batch generates until its
longest row is done) and through frame-level admission with 32 slots.  Single engine, whole chip, no lane pipelining."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import bench
from sopro_amd.continuous import ContinuousSynthesizer

tts, cfg, mc, wn, mn = bench.build_engine("cuda:0")
ids, ref_tq = bench.make_inputs(0)
ref = tts.prepare_reference(ref_tokens_tq=ref_tq)
rng = np.random.default_rng(5)
N = int(sys.argv[1]) if len(sys.argv) > 1 else 128
budgets = rng.integers(80, 401, size=N)
reqs = [dict(text_ids=ids[i % 32], ref=ref, max_frames=int(b) - 1, top_p=0.9, temperature=1.05, anti_loop=True) for i, b in enumerate(budgets)]
audio_s = float(budgets.sum()) * 0.08


def static():
    out = []
    for s in range(0, N, 32):
        chunk = reqs[s:s + 32]
        mf = max(r["max_frames"] for r in chunk)  # a static batch pads every row to its longest
        wav = tts.synthesize_batch([""] * len(chunk), [ref] * len(chunk), max_frames=mf, text_ids=[r["text_ids"] for r in chunk])
        out += [w[..., : (r["max_frames"] + 1) * 1920] for w, r in zip(wav, chunk)]
    return out


from sopro_amd.pipeline import PipelinedSynthesizer


def static_pipelined():
    pipe = PipelinedSynthesizer(tts, lanes=4, ar_cus=64, ar_parts=2, ar_shared=True)
    try:
        jobs = []
        for s in range(0, N, 32):
            chunk = reqs[s:s + 32]
            jobs.append(dict(texts=[""] * len(chunk), refs=[ref] * len(chunk), max_frames=max(r["max_frames"] for r in chunk),
                             text_ids=[r["text_ids"] for r in chunk]))
        for _ in range(2):
            pipe.run(jobs)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        pipe.run(jobs)
        torch.cuda.synchronize()
        return time.perf_counter() - t0
    finally:
        pipe.close()


dt = static_pipelined()
print(f"{'static batches, 4-lane pipeline':46s}: {dt * 1e3:8.1f} ms for {audio_s:.0f} audio-s -> {audio_s / dt:8.1f} audio-s/s", flush=True)
POLL = int(sys.argv[2]) if len(sys.argv) > 2 else 16


def timed(name, fn):
    fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    out = fn()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    assert all(o.shape[-1] == (r["max_frames"] + 1) * 1920 for o, r in zip(out, reqs))
    print(f"{name:46s}: {dt * 1e3:8.1f} ms for {audio_s:.0f} audio-s -> {audio_s / dt:8.1f} audio-s/s", flush=True)


timed("static batches of 32, one engine", static)
for name, kw in (("frame-level admission, 32 slots, one engine", dict(slots=32)),
                 ("admission, 2 x 32 slots on a 64-CU partition", dict(slots=32, ar_cus=64, generators=2))):
    eng = ContinuousSynthesizer(tts, max_frames=400, max_text=64, poll_every=POLL, bulk_batch=32, **kw)
    try:
        timed(name, lambda: eng.run(reqs))
        print("   stats", eng.stats, flush=True)
    finally:
        eng.close()
