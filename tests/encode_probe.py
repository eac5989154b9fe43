"""Time the reference-audio path on the GPU (Mimi encode of a 12 s reference, the reference's default ref_seconds)
and the CPU oracle beside it.  Usage: python tests/encode_probe.py [seconds]   (a probe, not a test: it lives here because only tests/ may import oracle/)"""
import json
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import sopro_oracle as O  # noqa: E402  (timing baseline only)
from sopro_amd.codec import MimiCodec  # noqa: E402
from sopro_amd.config import MimiDecoderConfig  # noqa: E402
from sopro_amd.weights import synth_mimi_weights  # noqa: E402

secs = float(sys.argv[1]) if len(sys.argv) > 1 else 12.0
mc = MimiDecoderConfig()
wnp = synth_mimi_weights(mc, 1234, with_encoder=True)
codec = MimiCodec(wnp, mc, device="cuda:0")
g = torch.Generator().manual_seed(0)
wav = (0.25 * torch.randn(int(secs * 24000), generator=g)).cuda()
for _ in range(3):
    codec.encode_waveform(wav)
torch.cuda.synchronize()
t0 = time.perf_counter()
n = 10
for _ in range(n):
    codes = codec.encode_waveform(wav)
torch.cuda.synchronize()
gpu_ms = (time.perf_counter() - t0) / n * 1e3
torch.set_num_threads(8)
mw = O.to_torch(wnp)
t0 = time.perf_counter()
oc = O.mimi_encode(wav.cpu().view(1, 1, -1), mw, mc)[0].permute(1, 0)
cpu_ms = (time.perf_counter() - t0) * 1e3
print(json.dumps({"audio_s": secs, "frames": int(codes.shape[0]), "gpu_ms": round(gpu_ms, 3), "audio_s_per_s": round(secs / gpu_ms * 1e3, 1),
                  "cpu_oracle_ms_8thr": round(cpu_ms, 1), "codes_equal_frac": float((codes.cpu() == oc).float().mean())}))
