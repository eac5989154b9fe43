#!/usr/bin/env python
"""r06: the refinement alone (128 x 200 frames) on a CU-masked stream - the command rocprofv3 traces for tools/r06/nar_table.py.
    python tools/r06/nar_run.py [cus] [reps]"""
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
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 4
B = int(os.environ.get("NAR_ROWS", "128"))
torch.cuda.set_device(0)
tts, cfg, mc, wn, mn = bench.build_engine("cuda:0", "f32")
m = tts.model
rng = np.random.default_rng(0)
ids, _ = bench.make_inputs(0)
voices = [tts.prepare_reference(ref_tokens_tq=v) for v in bench.make_voices(0, 32)]
idsB, refsB = (ids * 4)[:B], (voices * 4)[:B]
prep = m.phase_cond(idsB, refsB, max_frames=199, style_strength=float(cfg.style_strength))
cond = prep["cond_ar"][:, :200].contiguous()
cb0 = torch.from_numpy(rng.integers(0, 2048, size=(B, 200)).astype(np.int32)).to("cuda:0")
if cus < 256:
    m.bulk_stream = hip.cu_range_stream(256 - cus, cus, torch.device("cuda:0"))
if os.environ.get("NAR_EAGER", "0") == "1":
    m.use_graph = False
for _ in range(3):
    t = m._nar_pass(cond, cond.stride(0), cb0, cb0.stride(0), [200] * B, B, 200, sync=True, raw=True)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(reps):
    t = m._nar_pass(cond, cond.stride(0), cb0, cb0.stride(0), [200] * B, B, 200, sync=True, raw=True)
torch.cuda.synchronize()
print(f"refinement {B} x 200 on {cus} CUs: {(time.perf_counter() - t0) / reps * 1e3:.3f} ms per pass; checksum {int(t.long().sum())}")
