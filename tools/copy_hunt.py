#!/usr/bin/env python
"""Developer probe (round 3): which host lines launch torch's own copy / fill kernels in a pipelined pass (they are not part of
the library's sequences).  Runs a few coalesced passes under torch.profiler and prints the Python source lines that launched
`elementwise_kernel*` / `FillFunctor` kernels, with counts per pass."""
import collections
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from torch.profiler import ProfilerActivity, profile

import bench
from sopro_amd.pipeline import PipelinedSynthesizer

tts, cfg, mc, wn, mn = bench.build_engine("cuda:0")
ids = bench.make_inputs(0)[0]
refs = [tts.prepare_reference(ref_tokens_tq=t) for t in bench.make_voices(0, 32)]
job = dict(texts=[""] * 32, refs=list(refs), max_frames=199, top_p=0.9, temperature=1.05, anti_loop=True, text_ids=list(ids), seed=1)
job64 = dict(job, texts=[""] * 64, refs=list(refs) * 2, text_ids=list(ids) * 2)  # one coalesced pass, on the calling thread (the profiler sees it)
for _ in range(3):
    tts.synthesize_batch(**job64)
torch.cuda.synchronize()
import traceback

sites = collections.Counter()


def wrap(obj, name):
    orig = getattr(obj, name)

    def f(*a, **k):
        t = a[0] if a and isinstance(a[0], torch.Tensor) else None
        if t is None or t.is_cuda or any(isinstance(x, torch.Tensor) and x.is_cuda for x in a[1:2]):
            fr = [x for x in traceback.extract_stack()[:-1] if "sopro_amd" in x.filename]
            if fr:
                sites[(name, f"{os.path.basename(fr[-1].filename)}:{fr[-1].lineno}")] += 1
        return orig(*a, **k)

    setattr(obj, name, f)


for nm in ("copy_", "contiguous", "to", "clone", "zero_", "fill_", "index_select", "clamp", "long", "float", "__setitem__"):
    wrap(torch.Tensor, nm)
for nm in ("stack", "cat", "zeros", "tensor"):
    wrap(torch, nm)
N = 4
for _ in range(N // 2):
    tts.synthesize_batch(**job64)
torch.cuda.synchronize()
for (nm, where), n in sites.most_common(40):
    print(f"{n / (N / 2):7.1f} per pass  {nm:14s} {where}")
