"""Generate the golden fixtures under tests/golden/ from THE REFERENCE ITSELF.

Runs only in the build container (needs /root/reference and the installed HuggingFace
transformers Mimi implementation); the GPU box never runs it.  It

  1. builds the reference object graph without ``from_pretrained`` (SURVEY.md 8c recipe),
  2. loads the seeded synthetic checkpoints of ``sopro_amd.weights`` into it with
     ``strict=True`` (which also proves our layout table matches the reference's),
  3. runs the reference stage by stage on seeded inputs and stores inputs + outputs,
  4. runs the oracle on the same inputs and prints the deviation (the committed test
     ``tests/test_oracle_golden.py`` re-checks this from the stored files).

Usage:  python tests/golden/make_golden.py
"""
from __future__ import annotations

import os
import sys

sys.dont_write_bytecode = True
HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, "/root/reference/src")

import numpy as np
import torch

from oracle import sopro_oracle as O
from sopro_amd.config import MimiDecoderConfig, SoproTTSConfig
from sopro_amd.weights import synth_mimi_weights, synth_sopro_weights

VOCAB = 512
SEED = 1234
torch.set_num_threads(4)


class FakeTok:
    vocab_size = VOCAB

    def __init__(self):
        self.table = {}

    def encode(self, text):
        return list(self.table[text])


def build_reference(weights, mimi_weights, cfg):
    from sopro.config import SoproTTSConfig as RefCfg
    from sopro.model import SoproTTS, SoproTTSModel
    from sopro.codec.mimi import MimiCodec
    from transformers import MimiConfig, MimiModel

    rcfg = RefCfg()
    tok = FakeTok()
    model = SoproTTSModel(rcfg, tok).eval()
    sd = {k: torch.from_numpy(v) for k, v in weights.items()}
    model.load_state_dict(sd, strict=True)
    codec = MimiCodec.__new__(MimiCodec)
    codec.device = torch.device("cpu")
    mm = MimiModel(MimiConfig(num_quantizers=32)).eval()
    missing, unexpected = mm.load_state_dict({k: torch.from_numpy(v) for k, v in mimi_weights.items()}, strict=False)
    assert not unexpected, unexpected
    assert all(k.startswith(("encoder", "downsample", "quantizer")) for k in missing), [k for k in missing if not k.startswith(("encoder", "downsample", "quantizer"))][:5]
    assert not [k for k in missing if "output_proj" in k or "embed_sum" in k or "cluster_usage" in k]
    for m in mm.modules():  # lazily cached codebooks must be rebuilt from the loaded buffers
        if hasattr(m, "_embed"):
            m._embed = None
    codec.model = mm
    tts = SoproTTS(model, rcfg, tok, codec, "cpu")
    return tts, tok


def maxdiff(a, b):
    a = a.detach().float() if isinstance(a, torch.Tensor) else torch.as_tensor(a).float()
    b = b.detach().float() if isinstance(b, torch.Tensor) else torch.as_tensor(b).float()
    return float((a - b).abs().max())


def main():
    cfg = SoproTTSConfig()
    mc = MimiDecoderConfig()
    weights = synth_sopro_weights(cfg, VOCAB, SEED)
    mweights = synth_mimi_weights(mc, SEED)
    tts, tok = build_reference(weights, mweights, cfg)
    model = tts.model
    w = O.to_torch(weights)
    mw = O.to_torch(mweights)
    rng = np.random.default_rng(SEED)
    out = {}

    # ---- prepare_reference / prepare_conditioning -------------------------------
    S, TR, MAXF = 19, 30, 48
    ids = rng.integers(0, VOCAB, size=S).astype(np.int64)
    ref_tq = rng.integers(0, 2048, size=(TR, 32)).astype(np.int64)
    with torch.inference_mode():
        pref = tts.prepare_reference(ref_tokens_tq=torch.from_numpy(ref_tq))
        prep = model.prepare_conditioning(torch.from_numpy(ids), pref, max_frames=MAXF, device=torch.device("cpu"), style_strength=1.2)
    oref = O.prepare_reference(torch.from_numpy(ref_tq), w, cfg)
    oprep = O.prepare_conditioning(torch.from_numpy(ids), oref, w, cfg, max_frames=MAXF, style_strength=1.2)
    print("prepare_reference: sv", maxdiff(pref.sv_ref, oref.sv_ref), "ref_seq", maxdiff(pref.ref_seq, oref.ref_seq),
          "k0", maxdiff(pref.ref_kv_caches[0]["k"], oref.ref_kv_caches[0]["k"]), "v2", maxdiff(pref.ref_kv_caches[2]["v"], oref.ref_kv_caches[2]["v"]))
    print("prepare_conditioning: txt_seq", maxdiff(prep["txt_seq"], oprep["txt_seq"]), "cond_ar", maxdiff(prep["cond_ar"], oprep["cond_ar"]))
    np.savez_compressed(os.path.join(HERE, "prep.npz"), ids=ids, ref_tq=ref_tq, max_frames=MAXF, style_strength=1.2,
                        sv_ref=pref.sv_ref.numpy(), ref_seq=pref.ref_seq.numpy(),
                        ref_k0=pref.ref_kv_caches[0]["k"].numpy(), ref_v2=pref.ref_kv_caches[2]["v"].numpy(),
                        txt_seq=prep["txt_seq"].numpy(), txt_pool=prep["txt_pool"].numpy(), cond_ar=prep["cond_ar"].numpy())

    # ---- AR teacher-forced logits, B=2, ragged text mask ------------------------
    T, S2 = 40, 17
    x = torch.from_numpy(rng.standard_normal((2, T, 384)).astype(np.float32))
    txt = torch.from_numpy(rng.standard_normal((2, S2, 384)).astype(np.float32))
    mask = torch.ones(2, S2, dtype=torch.bool)
    mask[1, 11:] = False
    with torch.inference_mode():
        ref_par = model.ar(x, text_emb=txt, text_mask=mask)
        st = model.ar.init_stream_state(2, torch.device("cpu"), torch.float32, text_emb=txt, text_mask=mask)
        steps = []
        for t in range(T):
            lg, st = model.ar.step(x[:, t:t + 1], st, text_emb=txt, text_mask=mask)
            steps.append(lg[:, 0])
        ref_step = torch.stack(steps, dim=1)
    ost = O.ar_init_state(2, txt, mask, w, cfg)
    o_step = torch.stack([O.ar_step(x[:, t], ost, w, cfg) for t in range(T)], dim=1)
    o_par = O.ar_forward_teacher(x, txt, mask, w, cfg)
    print("ar teacher: ref step-vs-parallel", maxdiff(ref_par, ref_step), "oracle step", maxdiff(ref_step, o_step), "oracle parallel", maxdiff(ref_par, o_par))
    np.savez_compressed(os.path.join(HERE, "ar_teacher.npz"), x=x.numpy(), txt=txt.numpy(), mask=mask.numpy(),
                        logits=ref_step.numpy().astype(np.float32))

    # ---- greedy AR token list (top_p=0, anti_loop=False) ------------------------
    with torch.inference_mode():
        toks = [tk for _t, tk, _e in model.ar_stream(prep, max_frames=MAXF, top_p=0.0, temperature=1.0, anti_loop=False)]
    lg_list = []
    otoks = [tk for _t, tk, _e in O.ar_generate(oprep, w, cfg, max_frames=MAXF, top_p=0.0, temperature=1.0, anti_loop=False, collect_logits=lg_list)]
    margins = []
    hist = []
    for lg, tk in zip(lg_list, otoks):
        xs = O.penalised_logits(lg, hist, 1.0, 1.1)
        top2 = torch.topk(xs, 2).values
        margins.append(float(top2[0] - top2[1]))
        hist.append(tk)
    print("greedy tokens equal:", toks == otoks, "n", len(toks), "min top1-top2 margin", min(margins))
    np.savez_compressed(os.path.join(HERE, "ar_greedy.npz"), tokens=np.array(toks, dtype=np.int64), min_margin=min(margins),
                        logits_first8=torch.stack(lg_list[:8]).numpy())

    # greedy with temperature and an EOS-friendly head bias: exercises EOS / min_gen rules
    w_eos = dict(weights)
    hb = weights["ar.head.bias"].copy()
    hb[2048] = 3.9
    w_eos["ar.head.bias"] = hb
    model.ar.head.bias.data.copy_(torch.from_numpy(hb))
    with torch.inference_mode():
        ev = [(t, tk, e) for t, tk, e in model.ar_stream(prep, max_frames=MAXF, top_p=0.0, temperature=0.8, anti_loop=False, min_gen_frames=20)]
        gtoks = model.generate_tokens(torch.from_numpy(ids), pref, max_frames=MAXF, device=torch.device("cpu"), top_p=0.0,
                                      temperature=0.8, anti_loop=False, style_strength=1.2, min_gen_frames=20)
    weos_t = O.to_torch(w_eos)
    oev = [(t, tk, e) for t, tk, e in O.ar_generate(oprep, weos_t, cfg, max_frames=MAXF, top_p=0.0, temperature=0.8, anti_loop=False, min_gen_frames=20)]
    ogt = O.generate_tokens(torch.from_numpy(ids), oref, weos_t, cfg, max_frames=MAXF, top_p=0.0, temperature=0.8,
                            anti_loop=False, style_strength=1.2, min_gen_frames=20)
    eos_steps = [t for t, tk, e in ev if e]
    print("eos run: steps", len(ev), "eos at", eos_steps, "equal", ev == oev, "generate_tokens T", tuple(gtoks.shape), "equal", torch.equal(gtoks, ogt))
    np.savez_compressed(os.path.join(HERE, "ar_eos.npz"), eos_bias=3.9, temperature=0.8, min_gen_frames=20,
                        tokens=np.array([tk for _t, tk, _e in ev], dtype=np.int64), gen_tokens=gtoks.numpy())
    model.ar.head.bias.data.copy_(torch.from_numpy(weights["ar.head.bias"]))

    # ---- NAR refinement ---------------------------------------------------------
    TN = 37
    cond = prep["cond_ar"][:, :TN]
    rvq1 = torch.from_numpy(rng.integers(0, 2048, size=(1, TN)).astype(np.int64))
    with torch.inference_mode():
        nar_ref = model.nar_refine(cond, rvq1)
    lgs = {}
    nar_o = O.nar_refine(cond, rvq1, w, cfg, collect_logits=lgs)
    nar_margin = min(float((torch.topk(v, 2).values[..., 0] - torch.topk(v, 2).values[..., 1]).min()) for v in lgs.values())
    print("nar tokens equal:", torch.equal(nar_ref, nar_o), "min margin", nar_margin)
    np.savez_compressed(os.path.join(HERE, "nar.npz"), T=TN, rvq1=rvq1.numpy(), tokens=nar_ref.numpy(), min_margin=nar_margin,
                        logits_cb1=lgs[1].numpy().astype(np.float32)[:, :8])

    # ---- Mimi decode ------------------------------------------------------------
    tok8 = rng.integers(0, 2048, size=(8, 32)).astype(np.int64)
    tok32 = rng.integers(0, 2048, size=(32, 32)).astype(np.int64)
    mm = tts.codec.model
    with torch.inference_mode():
        codes8 = torch.from_numpy(tok8).permute(1, 0).unsqueeze(0).contiguous()
        emb = mm.quantizer.decode(codes8)
        up = mm.upsample(emb)
        tr = mm.decoder_transformer(up.transpose(1, 2), return_dict=True)[0].transpose(1, 2)
        wav8 = tts.codec.decode_full(torch.from_numpy(tok8))
        wav32 = tts.codec.decode_full(torch.from_numpy(tok32))
    taps = {}
    o8 = O.mimi_decode(codes8, mw, mc, taps=taps)
    o32 = O.decode_full(torch.from_numpy(tok32), mw, mc)
    print("mimi: rvq", maxdiff(emb, taps["rvq"]), "up", maxdiff(up, taps["upsample"]), "tr", maxdiff(tr, taps["transformer"]),
          "wav8", maxdiff(wav8, o8), "wav32", maxdiff(wav32, o32), "|wav|max", float(wav32.abs().max()), "shape", tuple(wav32.shape))
    np.savez_compressed(os.path.join(HERE, "mimi.npz"), tok8=tok8, tok32=tok32, rvq=emb.numpy(), upsample=up.numpy(),
                        transformer=tr.numpy(), wav8=wav8.numpy(), wav32=wav32.numpy())

    # ---- Mimi encode (reference audio -> tokens) + host audio helpers ------------
    from transformers import MimiConfig, MimiModel
    from sopro.audio import center_crop_audio, trim_silence_energy

    rng_e = np.random.default_rng(4321)  # own stream: the fixtures above keep their draws
    mwe = synth_mimi_weights(mc, SEED, with_encoder=True)
    mme = MimiModel(MimiConfig(num_quantizers=int(mc.num_quantizers))).eval()
    print("mimi encoder load:", mme.load_state_dict({k: torch.from_numpy(v) for k, v in mwe.items()}, strict=True))
    N = 24000 + 777
    ewav = torch.from_numpy((0.3 * rng_e.standard_normal(N)).astype(np.float32)).view(1, 1, N)
    with torch.inference_mode():
        eemb = mme.encoder(ewav)
        etr = mme.encoder_transformer(eemb.transpose(1, 2), return_dict=True)[0].transpose(1, 2)
        eds = mme.downsample(etr)
        ecodes = mme.encode(ewav, return_dict=True).audio_codes
    etaps = {}
    ocodes = O.mimi_encode(ewav, O.to_torch(mwe), mc, etaps)
    print("mimi encode: seanet", maxdiff(eemb, etaps["enc_seanet"]), "tr", maxdiff(etr, etaps["enc_transformer"]), "ds",
          maxdiff(eds, etaps["enc_downsample"]), "codes equal", float((ecodes == ocodes).float().mean()), tuple(ecodes.shape))
    # silence | burst | silence at 16 kHz for the trim rule
    sr16 = 16000
    tw = 1e-4 * rng_e.standard_normal(int(1.6 * sr16)).astype(np.float32)
    tw[int(0.41 * sr16):int(1.23 * sr16)] += (0.2 * np.sin(np.arange(int(1.23 * sr16) - int(0.41 * sr16)) * 0.07)).astype(np.float32)
    trimmed = trim_silence_energy(torch.from_numpy(tw), sr16)
    cropped = center_crop_audio(torch.from_numpy(tw)[None], 9999)
    np.savez_compressed(os.path.join(HERE, "mimi_encode.npz"), wav=ewav.numpy()[0, 0], codes=ecodes[0].permute(1, 0).numpy(),
                        enc_seanet_head=eemb.numpy()[:, :, :4], enc_downsample=eds.numpy(), trim_in=tw, trim_sr=sr16,
                        trim_out=trimmed.numpy(), crop_out=cropped.numpy()[0])

    # ---- a PreparedReference cache file as the reference's demo server writes it (demo/server.py:69-112) ----
    from sopro.model import PreparedReference as RefPR

    gc = torch.Generator().manual_seed(5)
    cache = RefPR(ref_tokens_btq=torch.randint(0, 2048, (1, 7, 32), generator=gc), sv_ref=torch.randn(1, 384, generator=gc),
                  ref_seq=torch.randn(1, 7, 384, generator=gc),
                  ref_kv_caches=[{"k": torch.randn(1, 2, 7, 192, generator=gc), "v": torch.randn(1, 2, 7, 192, generator=gc),
                                  "key_padding_mask": None} for _ in range(3)])
    torch.save(cache, os.path.join(HERE, "ref_cache_reference.pt"))

    # ---- end-to-end synthesize + stream (greedy) --------------------------------
    tok.table["hello"] = ids.tolist()
    with torch.inference_mode():
        wav = tts.synthesize("hello", ref=pref, max_frames=20, top_p=0.0, temperature=1.0, anti_loop=False, style_strength=1.2)
        chunks = list(tts.stream("hello", ref=pref, max_frames=20, top_p=0.0, temperature=1.0, anti_loop=False, style_strength=1.2, chunk_frames=6))
    owav = O.synthesize(torch.from_numpy(ids), oref, w, mw, cfg, mc, max_frames=20, top_p=0.0, temperature=1.0, anti_loop=False, style_strength=1.2)
    ochunks = list(O.stream(torch.from_numpy(ids), oref, w, mw, cfg, mc, max_frames=20, top_p=0.0, temperature=1.0, anti_loop=False, style_strength=1.2, chunk_frames=6))
    print("synthesize:", tuple(wav.shape), maxdiff(wav, owav), "stream chunks", [tuple(c.shape) for c in chunks],
          "oracle", [tuple(c.shape) for c in ochunks], "diff", max(maxdiff(a, b) for a, b in zip(chunks, ochunks)))
    np.savez_compressed(os.path.join(HERE, "e2e.npz"), max_frames=20, wav=wav.numpy(), chunk_sizes=np.array([c.shape[1] for c in chunks]),
                        stream=torch.cat(chunks, dim=1).numpy())
    print("fixtures written to", HERE)


if __name__ == "__main__":
    main()
