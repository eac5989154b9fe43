"""stream() fixtures at other chunk sizes than the reference default, generated from THE REFERENCE ITSELF
(build container only: needs /root/reference + the installed HF Mimi), same inputs as make_golden_full.py:

    python tests/golden/make_golden_stream_chunks.py

Writes stream_c1.npz (chunk_frames = 1, 48 frames: a refinement + decode per frame) and stream_c16.npz
(chunk_frames = 16, 96 frames): chunk sizes + concatenated samples.  The oracle runs next to the reference and
the deviation is printed; tests/test_oracle_full_size.py re-checks oracle == fixture on the CPU and
tests/test_gpu_full_size.py engine == fixture on the GPU.
"""
from __future__ import annotations

import os
import sys

sys.dont_write_bytecode = True
HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, HERE)
sys.path.insert(0, "/root/reference/src")

import numpy as np
import torch

from make_golden import SEED, VOCAB, build_reference, maxdiff
from make_golden_full import GREEDY, full_inputs
from oracle import sopro_oracle as O
from sopro_amd.config import MimiDecoderConfig, SoproTTSConfig
from sopro_amd.weights import synth_mimi_weights, synth_sopro_weights


def main():
    torch.set_num_threads(8)
    cfg, mc = SoproTTSConfig(), MimiDecoderConfig()
    weights = synth_sopro_weights(cfg, VOCAB, SEED, suppress_eos=True)
    mweights = synth_mimi_weights(mc, SEED)
    tts, tok = build_reference(weights, mweights, cfg)
    w, mw = O.to_torch(weights), O.to_torch(mweights)
    ids, ref_tq = full_inputs()
    tok.table["full"] = ids.tolist()
    with torch.inference_mode():
        pref = tts.prepare_reference(ref_tokens_tq=torch.from_numpy(ref_tq))
    oref = O.prepare_reference(torch.from_numpy(ref_tq), w, cfg)
    for name, cf, maxf in (("stream_c1", 1, 47), ("stream_c16", 16, 95)):
        with torch.inference_mode():
            chunks = list(tts.stream("full", ref=pref, max_frames=maxf, style_strength=1.0, chunk_frames=cf, **GREEDY))
        ochunks = list(O.stream(torch.from_numpy(ids), oref, w, mw, cfg, mc, max_frames=maxf, style_strength=1.0, chunk_frames=cf, **GREEDY))
        cat, ocat = torch.cat(chunks, dim=1), torch.cat(ochunks, dim=1)
        print(f"{name}: {len(chunks)} chunks, {cat.shape[1] // 1920} frames; oracle chunk sizes equal {[c.shape[1] for c in chunks] == [c.shape[1] for c in ochunks]}, "
              f"diff {maxdiff(cat, ocat):.3e} of |wav|max {float(cat.abs().max()):.3f}")
        np.savez_compressed(os.path.join(HERE, name + ".npz"), max_frames=maxf, chunk_frames=cf, chunk_sizes=np.array([c.shape[1] for c in chunks]),
                            stream=cat.numpy().reshape(-1))


if __name__ == "__main__":
    main()
