#!/usr/bin/env python
"""Developer probe (round 3, verdict item 1a): what does a dependent kernel boundary cost on the streams the AR frame uses?
Chains of 23 launches - (a) trivial one-workgroup kernels, (b) the FF1 skinny kernel (192 workgroups, its output unused by the
next link: only the stream order makes them dependent) - replayed from a hipGraph or launched eagerly, on an ordinary stream, a
CU-masked stream that allows every CU, and the 64-CU partition; one chain alone and two chains from two host threads.
Prints us per launch (wall time of many replays / launches)."""
import os
import sys
import threading
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from sopro_amd import hip

dev = torch.device("cuda:0")
hip.load()
N, REPS = 23, 300
tiny = [torch.zeros(64, device=dev) for _ in range(4)]
B, D = 32, 384
X = torch.randn(B, D, device=dev)
W1 = hip.pack_skinny_w(torch.randn(4 * D, D, device=dev) * 0.05)
b1 = torch.zeros(4 * D, device=dev)
U = [torch.empty(B, 4 * D, device=dev) for _ in range(2)]


def link(kind, i, lane):
    if kind == "trivial":
        hip.tanh_affine(tiny[2 * lane], tiny[2 * lane + 1], 0.0, 1.0, 64)
    else:
        hip.skinny(X, W1, U[lane], B=B, N=4 * D, K=D, rms_norm=True, bias=b1, epilogue=hip.EPI_GELU)


def chain(kind, mode, stream, lane, out, bar):
    with torch.cuda.stream(stream):
        link(kind, 0, lane)
        stream.synchronize()
        g = None
        if mode == "graph":
            hip.capture_begin()
            for i in range(N):
                link(kind, i, lane)
            g = hip.capture_end()
            g.launch()
            stream.synchronize()
        bar.wait()
        t0 = time.perf_counter()
        if g is not None:
            g.launch_n(REPS)
        else:
            for _ in range(REPS):
                for i in range(N):
                    link(kind, i, lane)
        stream.synchronize()
        out[lane] = (time.perf_counter() - t0) / (REPS * N) * 1e6


streams = {"ordinary": lambda: torch.cuda.Stream(), "mask-all-256": lambda: hip.cu_range_stream(0, 256, dev),
           "mask-64": lambda: hip.cu_range_stream(0, 64, dev)}
for kind in ("trivial", "ff1-192wg"):
    for sname, mk in streams.items():
        for mode in ("graph", "eager"):
            for lanes in (1, 2):
                ss = [mk() for _ in range(lanes)]
                res = []
                for rep in range(3):
                    out = [0.0] * lanes
                    bar = threading.Barrier(lanes)
                    th = [threading.Thread(target=chain, args=(kind, mode, ss[i], i, out, bar)) for i in range(lanes)]
                    for t in th:
                        t.start()
                    for t in th:
                        t.join()
                    res.append("/".join(f"{o:5.2f}" for o in out))
                print(f"{kind:10s} {sname:13s} {mode:5s} chains={lanes}: us per launch  " + "   ".join(res), flush=True)
