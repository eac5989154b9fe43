#!/usr/bin/env python
"""r06: the codec decoder alone (64 x 200 frames, recorded sequence) on a CU-masked stream, a few repeats - the command rocprofv3
traces for the per-launch table of tools/r06/decode_table.py.   python tools/r06/decode_run.py [cus] [reps]"""
import os
import sys
import time

R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, R)
import numpy as np  # noqa: E402
import torch  # noqa: E402

import bench  # noqa: E402
from sopro_amd import hip  # noqa: E402

cus = int(sys.argv[1]) if len(sys.argv) > 1 else 192
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 6
torch.cuda.set_device(0)
tts, cfg, mc, wn, mn = bench.build_engine("cuda:0", os.environ.get("DECODE_PRECISION", "f32"))
toks = torch.from_numpy(np.random.default_rng(0).integers(0, 2048, size=(64, 200, 32)).astype(np.int32)).to("cuda:0")
codec = tts.codec
if cus < 256:
    codec.stream = hip.cu_range_stream(256 - cus, cus, torch.device("cuda:0"))
if os.environ.get("DECODE_EAGER", "0") == "1":
    codec.use_graph = False
for _ in range(3):
    codec.decode_batch(toks)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(reps):
    w = codec.decode_batch(toks)
torch.cuda.synchronize()
print(f"decode 64 x 200 on {cus} CUs: {(time.perf_counter() - t0) / reps * 1e3:.3f} ms per decode; checksum {float(w.double().abs().sum()):.6e}")
