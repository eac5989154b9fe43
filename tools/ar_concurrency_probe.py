#!/usr/bin/env python
"""Developer probe: how well do two AR phases (recorded frame graphs of two engines) overlap on one GPU?
    python tools/ar_concurrency_probe.py [B] [steps]
Cases: one phase alone / two at once, on ordinary streams (whole chip), on one shared CU-masked partition of 64 / 96 CUs,
and on two disjoint partitions.  Prints us per frame of each phase (wall time of the pair / steps)."""
import os
import sys
import threading
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from bench import build_engine, make_inputs
from sopro_amd import hip
from sopro_amd.model import _ARRun

B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 200
tts, cfg, mc, wn, mn = build_engine("cuda:0")
dev = tts.device
ids, ref_tq = make_inputs(0)
ref = tts.prepare_reference(ref_tokens_tq=ref_tq)
NL = int(os.environ.get("PROBE_LANES", "2"))
lanes = [tts] + [tts.clone_lane() for _ in range(NL - 1)]
kw = dict(top_p=0.9, temperature=1.05, anti_loop=True)
preps = [l.model.prepare_conditioning_batch(ids[:B], [ref] * B, max_frames=steps - 1) for l in lanes]
torch.cuda.synchronize()


def phase(lane, prep, out, i):
    with torch.cuda.stream(lane.model.stream):
        run = _ARRun(lane.model, prep["cond_ar"], prep["txt_seq"], prep["text_lens"], min_gen_frames=None, **kw)
        lane.model.stream.synchronize()
        bar.wait()
        t0 = time.perf_counter()
        run.advance(steps)
        host[i] = (time.perf_counter() - t0) / steps * 1e6  # the enqueue alone (the launch thread's cost per frame)
        lane.model.stream.synchronize()
        out[i] = (time.perf_counter() - t0) / steps * 1e6


def case(name, streams):
    global bar, host
    n = len(streams)
    for l, s in zip(lanes, streams):
        l.model.stream = l.model.prep_stream = l.model.bulk_stream = s
        l.model._ar_cache.clear()
    res = []
    for rep in range(3):
        out = [0.0] * n
        host = [0.0] * n
        bar = threading.Barrier(n)
        th = [threading.Thread(target=phase, args=(lanes[i], preps[i], out, i)) for i in range(n)]
        for t in th:
            t.start()
        for t in th:
            t.join()
        res.append("/".join(f"{o:6.1f}" for o in out) + " (host " + "/".join(f"{h:.0f}" for h in host) + ")")
    print(f"{name:44s} us/frame per phase: " + "   ".join(res), flush=True)


S = lambda: torch.cuda.Stream(device=dev)  # noqa: E731
M = lambda lo, n: hip.cu_range_stream(lo, n, dev)  # noqa: E731
if len(sys.argv) > 3 and sys.argv[3] == "masks":
    # mask bit i = XCD i % 8, CU slot j = i // 8 of that XCD: which slots share a shader engine?
    def K(pred):
        bits = [i for i in range(256) if pred(i // 8)]
        return lambda: hip.cu_mask_stream(bits, dev), len(bits)
    layouts = {"slots 0-7 (range)": K(lambda j: j < 8), "slots j % 4 == 0": K(lambda j: j % 4 == 0),
               "slots j % 2 == 0, j < 16": K(lambda j: j % 2 == 0 and j < 16), "slots j % 8 == 0 or 1": K(lambda j: j % 8 < 2),
               "slots 0-11 (range)": K(lambda j: j < 12), "slots j % 4 == 0 + j % 8 == 1": K(lambda j: j % 4 == 0 or j % 8 == 1),
               "slots j % 2 == 0": K(lambda j: j % 2 == 0), "slots 0-15 (range)": K(lambda j: j < 16)}
    for name, (mk, n) in layouts.items():
        case(f"1 phase, {n} CUs, {name}", [mk()])
        case(f"2 phases shared, {n} CUs, {name}", [mk(), mk()])
    sys.exit(0)
if NL > 2:  # PROBE_LANES=4 python tools/ar_concurrency_probe.py 16 200: more, smaller chains on the shared partition
    for n in range(1, NL + 1):
        case(f"{n} phases of {B} rows, one shared 64-CU partition", [M(0, 64) for _ in range(n)])
    sys.exit(0)
case("1 phase, whole chip", [S()])
case("2 phases, whole chip, ordinary streams", [S(), S()])
for n in (64, 96, 128):
    case(f"1 phase, {n}-CU partition", [M(0, n)])
    case(f"2 phases, one shared {n}-CU partition", [M(0, n), M(0, n)])
case("2 phases, disjoint 64 + 64 CUs", [M(0, 64), M(64, 64)])
case("2 phases: whole chip + 64-CU partition", [S(), M(0, 64)])
case("2 phases: 64-CU partition + whole chip", [M(0, 64), S()])
case("2 phases: whole chip + CUs 64..255", [S(), M(64, 192)])
case("2 phases, disjoint 128 + 128 CUs", [M(0, 128), M(128, 128)])
