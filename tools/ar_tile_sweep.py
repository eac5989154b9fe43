#!/usr/bin/env python
"""Developer probe (round 3): workgroup shapes of the AR-step kernels (sopro_skinny_args.mt x nt) on the generation partition.
For every shape spec: one phase alone on the whole chip, then two phases on the shared 64-CU partition with nothing / a copy loop
/ the decoder-sized contraction on the other 192 CUs.  Prints us per frame; the bytes a frame's workgroups ingest are printed
next to it (weights once per mt*16 rows, activations once per nt*16 columns).
    python tools/ar_tile_sweep.py ["spec;spec;..."]      spec = '2x2' or 'glu:2x1,ff1:2x2,ff2:2x2,head:2x2'"""
import os
import sys
import threading
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from bench import build_engine, make_inputs
from sopro_amd import hip
from sopro_amd.model import _ARRun

B, steps = int(os.environ.get("PROBE_B", 32)), 200
SPECS = (sys.argv[1] if len(sys.argv) > 1 else "1x1;2x1;1x2;2x2;glu:1x2,ff1:2x2,ff2:2x2,head:2x2;glu:1x1,ff1:2x2,ff2:2x2,head:1x2").split(";")
LOADS = os.environ.get("PROBE_LOADS", "none,stream,gemmbig").split(",")
tts, cfg, mc, wn, mn = build_engine("cuda:0")
dev = tts.device
ids, ref_tq = make_inputs(0)
ref = tts.prepare_reference(ref_tokens_tq=ref_tq)
lanes = [tts, tts.clone_lane()]
kw = dict(top_p=0.9, temperature=1.05, anti_loop=True)
idsB = [ids[i % len(ids)] for i in range(B)]
preps = [l.model.prepare_conditioning_batch(idsB, [ref] * B, max_frames=steps - 1) for l in lanes]
part = [hip.cu_range_stream(0, 64, dev) for _ in lanes]
whole = torch.cuda.Stream()
bulk = hip.cu_range_stream(64, 192, dev)
big_a, big_b = torch.empty(1 << 28, device=dev), torch.empty(1 << 28, device=dev)
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
                elif kind == "gemmbig":
                    hip.gemm(A2, W2, C2, M=102400, N=1536, K=1024)
            n += 1
            if n % 4 == 0:
                bulk.synchronize()
        bulk.synchronize()


def ingest_mb(tiles):
    """Bytes the workgroups of one 32-row frame load (weights + activation rows + partial sums + folded K'/V'), MB."""
    D, S, H = 384, 64, 4
    tot = 0.0
    def wg(kind, ntile_cols, n_tiles, kslices, np_, wcols):
        mt, nt = (int(v) for v in tiles[kind].split("x"))
        groups = -(-B // (16 * mt))
        wgs = -(-n_tiles // nt) * kslices * groups
        per = nt * wcols * D * 4 + mt * 16 * D * 4 * (1 + np_)
        return wgs * per
    for i in range(6):
        tot += wg("glu", 8, 48, 1, 3 if i > 0 else 0, 16)
        tot += wg("ff1", 16, 96, 1, 0, 16)
        tot += wg("ff2", 16, 24, 4, 0, 16)
    tot += wg("head", 16, 129, 1, 3, 16)
    tot += 3 * B * H * 2 * S * D * 4
    return tot / 1e6


def set_stream(l, s):
    l.model.stream = l.model.prep_stream = l.model.bulk_stream = s
    l.model._ar_cache.clear()


def phase(lane, prep, out, i, bar):
    with torch.cuda.stream(lane.model.stream):
        run = _ARRun(lane.model, prep["cond_ar"], prep["txt_seq"], prep["text_lens"], min_gen_frames=None, **kw)
        run.advance(8)
        lane.model.stream.synchronize()
        bar.wait()
        t0 = time.perf_counter()
        run.advance(steps - 8)
        lane.model.stream.synchronize()
        out[i] = (time.perf_counter() - t0) / (steps - 8) * 1e6


def measure(n_lanes):
    out = [0.0] * n_lanes
    bar = threading.Barrier(n_lanes)
    th = [threading.Thread(target=phase, args=(lanes[i], preps[i], out, i, bar)) for i in range(n_lanes)]
    for t in th:
        t.start()
    for t in th:
        t.join()
    return out


for spec in SPECS:
    for l in lanes:
        l.model.set_ar_tiles(spec)
    tiles = dict(lanes[0].model.ar_tiles)
    set_stream(lanes[0], whole)
    alone = min(measure(1)[0] for _ in range(3))
    for l, s in zip(lanes, part):
        set_stream(l, s)
    one64 = min(measure(1)[0] for _ in range(2))
    line = f"{spec:44s} ingest {ingest_mb(tiles):6.1f} MB/frame | whole chip 1 phase {alone:6.1f} | 64 CUs 1 phase {one64:6.1f} | 64 CUs 2 phases:"
    for kind in LOADS:
        stop.clear()
        bg = None
        if kind != "none":
            bg = threading.Thread(target=background, args=(kind,))
            bg.start()
            time.sleep(0.3)
        res = [measure(2) for _ in range(3)]
        stop.set()
        if bg is not None:
            bg.join()
        best = min(res, key=lambda r: sum(r))
        line += f"  {kind} {best[0]:6.1f}/{best[1]:6.1f}"
    print(line, flush=True)
