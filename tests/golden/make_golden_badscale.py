"""Range-robustness fixture, generated from THE REFERENCE on badly scaled synthetic checkpoints
(sopro_amd.weights.badly_scaled_sopro / badly_scaled_mimi): NAR refinement of a conditioning block and a Mimi decode.  Every
parity fixture before this one used N(0, sigma) weights, which never leave the comfortable range of the f16 / bf16 piece
arithmetic; the reference itself is range-free fp32 (src/sopro/nn/blocks.py:26-37, nn/nar.py:89-116).

Runs only in the build container (needs /root/reference and the installed HuggingFace Mimi).  Usage:
    python tests/golden/make_golden_badscale.py
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

from make_golden import VOCAB, build_reference, maxdiff
from oracle import sopro_oracle as O
from sopro_amd.config import MimiDecoderConfig, SoproTTSConfig
from sopro_amd.weights import badly_scaled_mimi, badly_scaled_sopro, synth_mimi_weights, synth_sopro_weights

SEED = 4242
torch.set_num_threads(4)


def run(overflow: bool, name: str):
    cfg, mc = SoproTTSConfig(), MimiDecoderConfig()
    weights = badly_scaled_sopro(synth_sopro_weights(cfg, VOCAB, SEED), cfg, overflow=overflow)
    mweights = badly_scaled_mimi(synth_mimi_weights(mc, SEED), mc)
    tts, _tok = build_reference(weights, mweights, cfg)
    model = tts.model
    w, mw = O.to_torch(weights), O.to_torch(mweights)
    rng = np.random.default_rng(SEED)
    S, TR, T = 23, 40, 64
    ids = rng.integers(0, VOCAB, size=S).astype(np.int64)
    ref_tq = rng.integers(0, 2048, size=(TR, 32)).astype(np.int64)
    rvq1 = torch.from_numpy(rng.integers(0, 2048, size=(1, T)).astype(np.int64))
    with torch.inference_mode():
        pref = tts.prepare_reference(ref_tokens_tq=torch.from_numpy(ref_tq))
        prep = model.prepare_conditioning(torch.from_numpy(ids), pref, max_frames=T - 1, device=torch.device("cpu"), style_strength=1.0)
        cond = prep["cond_ar"][:, :T].contiguous()
        toks = model.nar_refine(cond, rvq1)
        wav = tts.codec.decode_full(toks[0]) if not overflow else None
    # the stream's scale, block by block, as the oracle sees it (what the f16 staging has to cope with)
    stats, orig = [], O.ssm_block

    def spy(x, w_, p, dil, causal):
        if p.startswith("nar."):
            r = x.pow(2).mean(-1).sqrt()
            stats.append((float(r.min()), float(r.max()), float(x.abs().max())))
        return orig(x, w_, p, dil, causal)

    O.ssm_block = spy
    lgs = {}
    try:
        otoks = O.nar_refine(cond, rvq1, w, cfg, collect_logits=lgs)
    finally:
        O.ssm_block = orig
    margin = min(float((torch.topk(v, 2).values[..., 0] - torch.topk(v, 2).values[..., 1]).min()) for v in lgs.values())
    lmax = max(float(v.abs().max()) for v in lgs.values())
    print(name, "nar tokens equal (reference vs oracle):", torch.equal(toks, otoks), "min logit margin", margin, "max |logit|", lmax)
    print("  residual stream rows over all stages / blocks: RMS %.3e ... %.3e, max |x| %.3e" % (min(s[0] for s in stats), max(s[1] for s in stats), max(s[2] for s in stats)))
    out = dict(seed=SEED, cond=cond.numpy(), rvq1=rvq1.numpy(), tokens=toks.numpy(), min_margin=margin, rel_margin=margin / lmax,
               stream_rms_min=min(s[0] for s in stats), stream_rms_max=max(s[1] for s in stats), stream_abs_max=max(s[2] for s in stats))
    if wav is not None:
        owav = O.decode_full(toks[0], mw, mc)
        print("  wav", tuple(wav.shape), "peak", float(wav.abs().max()), "oracle diff", maxdiff(wav, owav))
        out["wav"] = wav.numpy().reshape(-1)
    np.savez_compressed(os.path.join(HERE, name + ".npz"), **out)
    print("  written", os.path.join(HERE, name + ".npz"))


def main():
    run(False, "badscale")
    run(True, "badscale_overflow")


if __name__ == "__main__":
    main()
