"""Full-size golden fixtures, generated from THE REFERENCE ITSELF at the shapes bench.py quotes
(BASELINE.json configs[1] / configs[4]): S = 64 text tokens, a 150-frame reference voice, 200 and 400
generated frames, and a stream() run long enough for the codec transformer's cache to pass its
250-position window.  Build container only (needs /root/reference + the installed HF Mimi).

    python tests/golden/make_golden_full.py

Writes (next to this file):
  full200.npz   ids, ref_tq, AR tokens, NAR tokens [200, 32], waveform [384000], oracle decision margins
  full400.npz   the same at max_frames = 399 (401 AR steps run, 400 frames kept: EOS is suppressed)
  stream160.npz stream(chunk_frames=6), 160 frames: chunk sizes + concatenated samples
  stream_legacy.npz  the same policy with the reference's ``drop_cache_tail`` *legacy branch* active
                (src/sopro/codec/mimi.py:92-103): the installed transformers 5.x has no
                ``to_legacy_cache`` / ``from_legacy_cache``, so the script adds the two methods of the
                lock-pinned 4.57.6 API to ``DynamicCache`` for that run only (plain per-layer (k, v)
                tuples out, a DynamicCache of plain layers in); the trimming itself is executed by the
                reference's own code.
The oracle is run next to the reference on every fixture and the deviation printed; the committed
tests re-check oracle == fixture on the CPU and engine == fixture on the GPU.
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
from oracle import sopro_oracle as O
from sopro_amd.config import MimiDecoderConfig, SoproTTSConfig
from sopro_amd.weights import synth_mimi_weights, synth_sopro_weights

S_FULL, TR_FULL = 64, 150
GREEDY = dict(top_p=0.0, temperature=1.0, anti_loop=False)


def full_inputs():
    rng = np.random.default_rng(2024)
    ids = rng.integers(0, VOCAB, size=S_FULL).astype(np.int64)
    ref_tq = rng.integers(0, 2048, size=(TR_FULL, 32)).astype(np.int64)
    return ids, ref_tq


def margins_ar(lg_list, toks):
    out, hist = [], []
    for lg, tk in zip(lg_list, toks):
        xs = O.penalised_logits(lg, hist, 1.0, 1.1)
        top2 = torch.topk(xs, 2).values
        out.append(float(top2[0] - top2[1]))
        hist.append(tk)
    return np.asarray(out, dtype=np.float32)


def margins_nar(lgs, T, Q):
    m = np.full((T, Q), np.inf, dtype=np.float32)
    for cb, v in lgs.items():
        t2 = torch.topk(v[0], 2, dim=-1).values
        m[:, cb] = (t2[:, 0] - t2[:, 1]).numpy()
    return m


def install_legacy_cache_api():
    """The two methods of transformers 4.57.6's DynamicCache the reference's drop_cache_tail looks for."""
    from transformers.cache_utils import DynamicCache

    def to_legacy_cache(self):
        return tuple((layer.keys, layer.values) for layer in self.layers)

    @classmethod
    def from_legacy_cache(cls, past):
        return cls(ddp_cache_data=past)  # plain DynamicLayers holding the given tensors

    DynamicCache.to_legacy_cache = to_legacy_cache
    DynamicCache.from_legacy_cache = from_legacy_cache

    def remove():
        del DynamicCache.to_legacy_cache
        del DynamicCache.from_legacy_cache

    return remove


def main():
    torch.set_num_threads(8)
    cfg, mc = SoproTTSConfig(), MimiDecoderConfig()
    weights = synth_sopro_weights(cfg, VOCAB, SEED, suppress_eos=True)
    mweights = synth_mimi_weights(mc, SEED)
    tts, tok = build_reference(weights, mweights, cfg)
    model = tts.model
    w, mw = O.to_torch(weights), O.to_torch(mweights)
    ids, ref_tq = full_inputs()
    tok.table["full"] = ids.tolist()
    with torch.inference_mode():
        pref = tts.prepare_reference(ref_tokens_tq=torch.from_numpy(ref_tq))
    oref = O.prepare_reference(torch.from_numpy(ref_tq), w, cfg)

    for name, maxf in (("full200", 199), ("full400", 399)):
        with torch.inference_mode():
            toks = model.generate_tokens(torch.from_numpy(ids), pref, max_frames=maxf, device=torch.device("cpu"),
                                         style_strength=1.0, **GREEDY)
            wav = tts.codec.decode_full(toks)
        T = int(toks.shape[0])
        oprep = O.prepare_conditioning(torch.from_numpy(ids), oref, w, cfg, max_frames=maxf, style_strength=1.0)
        lg_list = []
        otoks_a = [tk for _t, tk, _e in O.ar_generate(oprep, w, cfg, max_frames=maxf, collect_logits=lg_list, **GREEDY)]
        lgs = {}
        otoks = O.nar_refine(oprep["cond_ar"][:, :T], torch.tensor(otoks_a[:T]).unsqueeze(0), w, cfg, collect_logits=lgs)[0]
        owav = O.decode_full(otoks, mw, mc)
        m_ar, m_nar = margins_ar(lg_list, otoks_a), margins_nar(lgs, T, 32)
        print(f"{name}: T={T} AR equal {toks[:, 0].tolist() == otoks_a[:T]} (oracle ran {len(otoks_a)} steps), NAR mismatches "
              f"{int((toks != otoks).sum())}, wav diff {maxdiff(wav, owav):.3e} of |wav|max {float(wav.abs().max()):.3f}, "
              f"min AR margin {m_ar.min():.3e}, min NAR margin {m_nar.min():.3e}")
        np.savez_compressed(os.path.join(HERE, name + ".npz"), ids=ids, ref_tq=ref_tq, max_frames=maxf, tokens=toks.numpy().astype(np.int16),
                            wav=wav.numpy().reshape(-1), ar_margin=m_ar, nar_margin=m_nar)

    def run_stream(maxf):
        with torch.inference_mode():
            return list(tts.stream("full", ref=pref, max_frames=maxf, style_strength=1.0, chunk_frames=6, **GREEDY))

    chunks = run_stream(159)
    ochunks = list(O.stream(torch.from_numpy(ids), oref, w, mw, cfg, mc, max_frames=159, style_strength=1.0, chunk_frames=6, **GREEDY))
    cat, ocat = torch.cat(chunks, dim=1), torch.cat(ochunks, dim=1)
    print("stream160:", len(chunks), "chunks", cat.shape[1] // 1920, "frames; oracle diff", maxdiff(cat, ocat), "of", float(cat.abs().max()))
    np.savez_compressed(os.path.join(HERE, "stream160.npz"), max_frames=159, chunk_sizes=np.array([c.shape[1] for c in chunks]),
                        stream=cat.numpy().reshape(-1))

    remove = install_legacy_cache_api()
    try:
        lchunks = run_stream(47)
    finally:
        remove()
    plain = run_stream(47)
    lcat, pcat = torch.cat(lchunks, dim=1), torch.cat(plain, dim=1)
    olc = list(O.stream(torch.from_numpy(ids), oref, w, mw, cfg, mc, max_frames=47, style_strength=1.0, chunk_frames=6, trim="legacy", **GREEDY))
    print("stream_legacy:", len(lchunks), "chunks; differs from the untrimmed policy by", maxdiff(lcat, pcat), "; oracle(trim=legacy) diff",
          maxdiff(lcat, torch.cat(olc, dim=1)), "of", float(lcat.abs().max()))
    np.savez_compressed(os.path.join(HERE, "stream_legacy.npz"), max_frames=47, chunk_sizes=np.array([c.shape[1] for c in lchunks]),
                        stream=lcat.numpy().reshape(-1))


if __name__ == "__main__":
    main()
