"""Reference-generated fixture for ``SoproTTS.encode_speaker`` (/root/reference/src/sopro/model.py:457-475): the reference
facade's own method on a 180-frame token matrix under three crop policies (default 12 s = 150 frames, 4 s = 50 frames,
``ref_seconds=0`` = no crop) -> ``tests/golden/speaker.npz``.  Build container only (needs /root/reference).

Usage:  python tests/golden/make_golden_speaker.py
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

from make_golden import SEED, VOCAB, build_reference
from sopro_amd.config import MimiDecoderConfig, SoproTTSConfig
from sopro_amd.weights import synth_mimi_weights, synth_sopro_weights


def main():
    cfg, mc = SoproTTSConfig(), MimiDecoderConfig()
    tts, _tok = build_reference(synth_sopro_weights(cfg, VOCAB, SEED), synth_mimi_weights(mc, SEED), cfg)
    g = torch.Generator().manual_seed(5150)
    ref_tq = torch.randint(0, 2048, (180, 32), generator=g)
    out = {"ref_tq": ref_tq.numpy()}
    for name, secs in (("default", None), ("sec4", 4.0), ("nocrop", 0.0)):
        sv = tts.encode_speaker(ref_tokens_tq=ref_tq, ref_seconds=secs)
        out["sv_" + name] = sv.numpy()
        print(name, tuple(sv.shape), float(sv.norm()))
    # the same vector prepare_reference stores (reference model.py:151-170)
    pref = tts.prepare_reference(ref_tokens_tq=ref_tq)
    assert torch.equal(pref.sv_ref.squeeze(0), torch.from_numpy(out["sv_default"]))
    np.savez_compressed(os.path.join(HERE, "speaker.npz"), **out)


if __name__ == "__main__":
    main()
