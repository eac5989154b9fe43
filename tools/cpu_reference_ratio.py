#!/usr/bin/env python
"""How fast is bench.py's CPU baseline (the oracle, `cpu_baseline.kind = "port"`) next to THE REFERENCE on one host?

/root/reference does not exist on the GPU box, so bench.py times the oracle there.  This script runs in the build container
(the only place both exist): the reference objects (tests/golden/make_golden.build_reference: /root/reference/src/sopro +
HuggingFace MimiModel, same synthetic checkpoint as bench.py) and the oracle synthesize the SAME bench utterances (S = 64,
150-frame voice prepared outside the timed part, 200 frames, reference default sampling) one at a time, interleaved, on the
same threads.  Writes profiles/r04_cpu_reference_ratio.json; bench.py copies `oracle_over_reference` into
`cpu_baseline.reference_ratio` (stamped as replayed).  VERDICT r3 weak 9.

    python tools/cpu_reference_ratio.py [n_utts=4] [threads=8]
"""
import json
import os
import sys
import time

sys.dont_write_bytecode = True
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
sys.path.insert(0, "/root/reference/src")

import numpy as np
import torch

import bench
from make_golden import build_reference
from oracle import sopro_oracle as O
from sopro_amd.config import MimiDecoderConfig, SoproTTSConfig
from sopro_amd.weights import synth_mimi_weights, synth_sopro_weights


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 4
    threads = int(sys.argv[2]) if len(sys.argv) > 2 else min(os.cpu_count() or 1, 8)
    torch.set_num_threads(threads)
    cfg, mc = SoproTTSConfig(), MimiDecoderConfig()
    wn = synth_sopro_weights(cfg, bench.VOCAB, 0, suppress_eos=True)
    mn = synth_mimi_weights(mc, 0)
    tts, tok = build_reference(wn, mn, cfg)
    w, mw = O.to_torch(wn), O.to_torch(mn)
    ids, ref_tq = bench.make_inputs(0)
    kw = dict(max_frames=bench.FRAMES - 1, top_p=0.9, temperature=1.05, anti_loop=True)
    with torch.inference_mode():
        rref = tts.prepare_reference(ref_tokens_tq=ref_tq)
        oref = O.prepare_reference(ref_tq, w, cfg)
        for i in range(len(ids)):
            tok.table[f"u{i}"] = ids[i].tolist()
        tts.synthesize("u0", ref=rref, max_frames=7)  # warm-ups
        O.synthesize(ids[0], oref, w, mw, cfg, mc, max_frames=7, top_p=0.9, temperature=1.05, anti_loop=True)
        t_ref = t_or = 0.0
        f_ref = f_or = 0
        for i in range(n):
            torch.manual_seed(100 + i)
            t0 = time.perf_counter()
            wav = tts.synthesize(f"u{i}", ref=rref, **kw)
            t_ref += time.perf_counter() - t0
            f_ref += wav.shape[-1] // 1920
            torch.manual_seed(100 + i)
            t0 = time.perf_counter()
            wav = O.synthesize(ids[i], oref, w, mw, cfg, mc, **kw)
            t_or += time.perf_counter() - t0
            f_or += wav.shape[-1] // 1920
    ref_v, or_v = f_ref * bench.FRAME_SEC / t_ref, f_or * bench.FRAME_SEC / t_or
    out = {"what": "sequential synthesize() of bench.py's utterances (S = 64, Tr = 150, 200 frames, reference default sampling), the "
                   "reference (/root/reference/src/sopro + transformers MimiModel) and oracle/sopro_oracle.py interleaved on one host",
           "host": {"cpu": next((l.split(":", 1)[1].strip() for l in open("/proc/cpuinfo") if l.startswith("model name")), "?"),
                    "cores_visible": os.cpu_count(), "threads": threads, "where": "build container (no GPU)"},
           "utterances": n, "frames": [f_ref, f_or],
           "reference_audio_s_per_s": round(ref_v, 3), "oracle_audio_s_per_s": round(or_v, 3),
           "oracle_over_reference": round(or_v / ref_v, 4),
           "torch": torch.__version__, "command": "python tools/cpu_reference_ratio.py " + " ".join(sys.argv[1:])}
    print(json.dumps(out, indent=1))
    json.dump(out, open(os.path.join(ROOT, "profiles", "r04_cpu_reference_ratio.json"), "w"), indent=1)


if __name__ == "__main__":
    main()
