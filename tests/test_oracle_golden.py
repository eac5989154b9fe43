"""The oracle (oracle/sopro_oracle.py) against outputs of THE REFERENCE ITSELF, stored by
tests/golden/make_golden.py (which imports /root/reference/src/sopro and HF MimiModel in the build
container).  This is what pins the oracle; the GPU tests then compare the HIP engine with it."""
import numpy as np
import torch

from conftest import VOCAB, golden
from oracle import sopro_oracle as O

torch.set_num_threads(4)


def _t(a):
    return torch.from_numpy(np.asarray(a))


def test_prepare_reference_and_conditioning(cfg, w):
    g = golden("prep")
    ref = O.prepare_reference(_t(g["ref_tq"]), w, cfg)
    assert torch.allclose(ref.sv_ref, _t(g["sv_ref"]), atol=2e-6)
    assert torch.allclose(ref.ref_seq, _t(g["ref_seq"]), atol=2e-5)
    assert torch.allclose(ref.ref_kv_caches[0]["k"], _t(g["ref_k0"]), atol=2e-5)
    assert torch.allclose(ref.ref_kv_caches[2]["v"], _t(g["ref_v2"]), atol=2e-5)
    prep = O.prepare_conditioning(_t(g["ids"]), ref, w, cfg, max_frames=int(g["max_frames"]), style_strength=float(g["style_strength"]))
    assert torch.allclose(prep["txt_seq"], _t(g["txt_seq"]), atol=2e-5)
    assert torch.allclose(prep["txt_pool"], _t(g["txt_pool"]), atol=2e-5)
    assert torch.allclose(prep["cond_ar"], _t(g["cond_ar"]), atol=5e-5)


def test_ar_teacher_forced_logits(cfg, w):
    g = golden("ar_teacher")
    x, txt, mask = _t(g["x"]), _t(g["txt"]), _t(g["mask"])
    st = O.ar_init_state(2, txt, mask, w, cfg)
    steps = torch.stack([O.ar_step(x[:, t], st, w, cfg) for t in range(x.shape[1])], dim=1)
    assert float((steps - _t(g["logits"])).abs().max()) < 2e-4
    par = O.ar_forward_teacher(x, txt, mask, w, cfg)
    assert float((par - _t(g["logits"])).abs().max()) < 2e-4


def _prep(cfg, w):
    g = golden("prep")
    ref = O.prepare_reference(_t(g["ref_tq"]), w, cfg)
    prep = O.prepare_conditioning(_t(g["ids"]), ref, w, cfg, max_frames=int(g["max_frames"]), style_strength=float(g["style_strength"]))
    return g, ref, prep


def test_ar_greedy_tokens(cfg, w):
    g, _ref, prep = _prep(cfg, w)
    gg = golden("ar_greedy")
    lg = []
    toks = [tk for _t2, tk, _e in O.ar_generate(prep, w, cfg, max_frames=int(g["max_frames"]), top_p=0.0, temperature=1.0,
                                                 anti_loop=False, collect_logits=lg)]
    assert toks == gg["tokens"].tolist()
    assert float((torch.stack(lg[:8]) - _t(gg["logits_first8"])).abs().max()) < 2e-4


def test_ar_eos_rule_and_generate_tokens(cfg, w, sopro_np):
    g, ref, prep = _prep(cfg, w)
    ge = golden("ar_eos")
    w2 = dict(w)
    hb = w["ar.head.bias"].clone()
    hb[2048] = float(ge["eos_bias"])
    w2["ar.head.bias"] = hb
    ev = list(O.ar_generate(prep, w2, cfg, max_frames=int(g["max_frames"]), top_p=0.0, temperature=float(ge["temperature"]),
                            anti_loop=False, min_gen_frames=int(ge["min_gen_frames"])))
    assert [tk for _a, tk, _e in ev] == ge["tokens"].tolist()
    toks = O.generate_tokens(_t(g["ids"]), ref, w2, cfg, max_frames=int(g["max_frames"]), top_p=0.0,
                             temperature=float(ge["temperature"]), anti_loop=False, style_strength=float(g["style_strength"]),
                             min_gen_frames=int(ge["min_gen_frames"]))
    assert torch.equal(toks, _t(ge["gen_tokens"]))


def test_nar_refine(cfg, w):
    g, _ref, prep = _prep(cfg, w)
    gn = golden("nar")
    T = int(gn["T"])
    lgs = {}
    toks = O.nar_refine(prep["cond_ar"][:, :T], _t(gn["rvq1"]), w, cfg, collect_logits=lgs)
    assert torch.equal(toks, _t(gn["tokens"]))
    assert float((lgs[1][:, :8] - _t(gn["logits_cb1"])).abs().max()) < 2e-4


def test_mimi_decode(mc, mw):
    g = golden("mimi")
    codes8 = _t(g["tok8"]).permute(1, 0).unsqueeze(0).contiguous()
    taps = {}
    o8 = O.mimi_decode(codes8, mw, mc, taps=taps)
    assert torch.allclose(taps["rvq"], _t(g["rvq"]), atol=1e-4)
    assert torch.allclose(taps["upsample"], _t(g["upsample"]), atol=1e-4)
    assert torch.allclose(taps["transformer"], _t(g["transformer"]), atol=5e-4)
    scale = float(_t(g["wav32"]).abs().max())
    assert float((o8 - _t(g["wav8"])).abs().max()) < 1e-4 * scale
    o32 = O.decode_full(_t(g["tok32"]), mw, mc)
    assert float((o32 - _t(g["wav32"])).abs().max()) < 1e-4 * scale


def test_end_to_end_synthesize_and_stream(cfg, mc, w, mw):
    g, ref, _prep_ = _prep(cfg, w)
    ge = golden("e2e")
    kw = dict(max_frames=int(ge["max_frames"]), top_p=0.0, temperature=1.0, anti_loop=False, style_strength=float(g["style_strength"]))
    wav = O.synthesize(_t(g["ids"]), ref, w, mw, cfg, mc, **kw)
    ref_wav = _t(ge["wav"])
    scale = float(ref_wav.abs().max())
    assert wav.shape == ref_wav.shape
    assert float((wav - ref_wav).abs().max()) < 1e-4 * scale
    chunks = list(O.stream(_t(g["ids"]), ref, w, mw, cfg, mc, chunk_frames=6, **kw))
    assert [int(c.shape[1]) for c in chunks] == ge["chunk_sizes"].tolist()
    assert float((torch.cat(chunks, dim=1) - _t(ge["stream"])).abs().max()) < 1e-4 * scale


def test_repeated_tail_and_sampling_distribution():
    # reference: src/sopro/sampling.py:16-21 and :52-80 (known-answer cases worked by hand)
    assert O.repeated_tail([1, 2, 3, 1, 2, 3]) is True
    assert O.repeated_tail([1, 2, 3, 1, 2, 4]) is False
    assert O.repeated_tail([5, 5, 5, 5]) is False  # n starts at 3 and needs 2n <= L
    lg = torch.tensor([2.0, 1.0, 0.0, -1.0])
    sp, si, forced = O.sampling_distribution(lg, [], top_p=0.0, temperature=1.0)
    assert forced is None and si[0].item() == 0 and float(sp[0]) == 1.0 and float(sp[1:].sum()) == 0.0
    sp, si, _ = O.sampling_distribution(lg, [0], top_p=1.0, temperature=1.0, repetition_penalty=2.0)
    p = torch.softmax(torch.tensor([1.0, 1.0, 0.0, -1.0]), 0)
    assert torch.allclose(sp.sort(descending=True).values, p.sort(descending=True).values, atol=1e-6)


def test_sampling_distribution_matches_reference_draws():
    """The oracle's sampling distribution against 20 000 seeded draws of THE REFERENCE's sample_token per case
    (tests/golden/make_golden_sampler.py; src/sopro/sampling.py:24-93 with top_k 50 / repetition penalty 1.1 of
    src/sopro/model.py:289-290): nothing outside the oracle's kept set was ever drawn, every kept token was drawn at its
    probability (5 sigma + 0.002), and the overall chi-square is in range.  Pins the temperature -> penalty -> softmax ->
    top-k -> renormalise -> top-p (shift by one) -> renormalise order."""
    g = golden("sampler")
    n = int(g["n_draws"])
    for ci, (top_p, temp, top_k, _scale) in enumerate(g["cases"].tolist()):
        logits, hist = _t(g[f"logits{ci}"]), g[f"hist{ci}"].tolist()
        counts = _t(g[f"counts{ci}"]).double()
        sp, si, forced = O.sampling_distribution(logits, hist, top_p, temp, top_k=int(top_k), repetition_penalty=1.1)
        assert forced is None
        p = torch.zeros(logits.numel(), dtype=torch.float64)
        p[si] = sp.double()
        assert float(counts[p == 0].sum()) == 0.0, (ci, "the reference drew a token outside the oracle's kept set")
        assert int((p > 0).sum()) == int((counts > 0).sum()), (ci, int((p > 0).sum()), int((counts > 0).sum()))
        freq = counts / n
        tol = 5.0 * torch.sqrt(p * (1 - p) / n) + 0.002
        assert float(((freq - p).abs() - tol).max()) <= 0.0, (ci, int(((freq - p).abs() - tol).argmax()))
        k = p > 0
        chi2 = float((((counts[k] - n * p[k]) ** 2) / (n * p[k])).sum())
        dof = int(k.sum()) - 1
        assert chi2 < dof + 6.0 * (2.0 * dof) ** 0.5 + 10.0, (ci, chi2, dof)
        # a different order of the steps is told apart by these fixtures: the penalty applied AFTER the softmax / top-k would
        # keep another set or other weights
        sp2, si2, _ = O.sampling_distribution(logits, [], top_p, temp, top_k=int(top_k), repetition_penalty=1.1)
        p2 = torch.zeros_like(p)
        p2[si2] = sp2.double()
        assert float((p2 - p).abs().max()) > 0.01, ci


def test_token2sv_matches_reference_encode_speaker(cfg, w):
    """SoproTTS.encode_speaker of the reference (src/sopro/model.py:457-475; fixture: tests/golden/make_golden_speaker.py)
    under its three crop policies (center crop: src/sopro/sampling.py:8-13)."""
    g = golden("speaker")
    ref_tq = _t(g["ref_tq"])
    for name, win in (("default", 150), ("sec4", 50), ("nocrop", None)):
        r = ref_tq
        if win is not None and r.shape[0] > win:
            s0 = (r.shape[0] - win) // 2
            r = r[s0: s0 + win]
        sv = O.token2sv(r.unsqueeze(0), w, int(cfg.codebook_size))
        assert float((sv.squeeze(0) - _t(g["sv_" + name])).abs().max()) < 2e-6, name


def test_mimi_encode(mc):
    """Oracle encoder (SEANet -> transformer -> downsample -> RVQ) against HF MimiModel.encode on the stored waveform."""
    from sopro_amd.weights import synth_mimi_weights
    from conftest import SEED

    g = golden("mimi_encode")
    mwe = O.to_torch(synth_mimi_weights(mc, SEED, with_encoder=True))
    taps = {}
    codes = O.mimi_encode(_t(g["wav"]).view(1, 1, -1), mwe, mc, taps)
    assert torch.allclose(taps["enc_seanet"][:, :, :4], _t(g["enc_seanet_head"]), atol=1e-4)
    assert torch.allclose(taps["enc_downsample"], _t(g["enc_downsample"]), atol=5e-4)
    assert torch.equal(codes[0].permute(1, 0), _t(g["codes"]))


def test_audio_helpers_match_reference():
    """Host trim/crop rules (sopro_amd.audio) and their oracle restatements against the reference's outputs."""
    from sopro_amd import audio

    g = golden("mimi_encode")
    sr = int(g["trim_sr"])
    assert np.array_equal(audio.trim_silence_energy(g["trim_in"], sr), g["trim_out"])
    assert torch.equal(O.trim_silence_energy(_t(g["trim_in"]), sr), _t(g["trim_out"]))
    assert np.array_equal(audio.center_crop_audio(g["trim_in"], 9999), g["crop_out"])
    assert torch.equal(O.center_crop_audio(_t(g["trim_in"]), 9999), _t(g["crop_out"]))
    assert audio.trim_silence_energy(g["trim_in"][:100], sr).shape[0] == 100  # shorter than 0.1 s: untouched


def test_resample_bank_matches_oracle_restatement():
    """The polyphase bank applied by sopro_fir1_f32 == the oracle's conv1d restatement of torchaudio's resampler."""
    from sopro_amd import audio

    rng = np.random.default_rng(5)
    for sr_in, sr_out in ((16000, 24000), (44100, 24000), (48000, 24000), (22050, 24000)):
        x = rng.standard_normal(3000).astype(np.float32)
        bank, left, orig, new = audio.sinc_resample_bank(sr_in, sr_out)
        n = x.shape[0]
        n_blk = n // orig + 1
        xp = np.concatenate([np.zeros(left, np.float32), x, np.zeros(left + orig + bank.shape[1], np.float32)])
        y = np.stack([bank @ xp[i * orig:i * orig + bank.shape[1]] for i in range(n_blk)]).reshape(-1)[: -(-new * n // orig)]
        ref = O.sinc_resample(_t(x), sr_in, sr_out).numpy()
        assert y.shape == ref.shape
        assert np.abs(y - ref).max() < 1e-5


def test_load_audio_file_formats(tmp_path):
    import struct
    import wave

    from sopro_amd import audio

    rng = np.random.default_rng(9)
    x = (rng.uniform(-0.9, 0.9, size=(1000, 2))).astype(np.float32)
    p16 = str(tmp_path / "a16.wav")
    with wave.open(p16, "wb") as f:
        f.setnchannels(2), f.setsampwidth(2), f.setframerate(22050)
        f.writeframes((x * 32767).astype("<i2").tobytes())
    w16, sr = audio.load_audio_file(p16)
    assert sr == 22050 and w16.shape == (1000,)
    assert np.abs(w16 - ((x * 32767).astype(np.int16) / 32768.0).mean(axis=1)).max() < 1e-6
    pf = str(tmp_path / "af.wav")
    body = x[:, 0].astype("<f4").tobytes()
    with open(pf, "wb") as f:
        f.write(b"RIFF" + struct.pack("<I", 36 + len(body)) + b"WAVE" + b"fmt " + struct.pack("<IHHIIHH", 16, 3, 1, 24000, 96000, 4, 32))
        f.write(b"data" + struct.pack("<I", len(body)) + body)
    wf, sr = audio.load_audio_file(pf)
    assert sr == 24000 and np.array_equal(wf, x[:, 0])


def test_sinc_resample_derivation_pins_alignment_gain_and_length():
    """torchaudio is not installable here, so ``sinc_resample`` (the restatement of ``torchaudio.functional.resample``'s
    published algorithm: Hann-windowed sinc, lowpass_filter_width 6, rolloff 0.99, zero padding, output length
    ceil(n * new / orig)) is pinned by what that algorithm must do, independently of how it is coded:
      * a sinusoid well inside both pass bands comes out as the SAME sinusoid sampled at the new rate (this fixes the
        polyphase alignment - an off-by-one phase or tap shift shows up as an O(1) error -, the filter gain and the kernel);
      * DC gain 1, output length ceil(n * new / orig), identity at equal rates;
      * down-by-2 then up-by-2 returns a band-limited signal.
    Interior samples only (the zero padding at the edges is part of the algorithm)."""
    for sr_in, sr_out in ((16000, 24000), (44100, 24000), (48000, 24000), (22050, 24000), (8000, 24000)):
        n = 4000
        t_in = np.arange(n) / sr_in
        f0 = 0.11 * min(sr_in, sr_out)  # well below 0.99 * Nyquist of both rates
        x = np.sin(2 * np.pi * f0 * t_in + 0.3).astype(np.float32)
        y = O.sinc_resample(_t(x), sr_in, sr_out).numpy()
        g = np.gcd(sr_in, sr_out)
        orig, new = sr_in // g, sr_out // g
        assert y.shape[0] == -(-n * new // orig)
        t_out = np.arange(y.shape[0]) / sr_out
        want = np.sin(2 * np.pi * f0 * t_out + 0.3)
        edge = int(0.02 * sr_out)  # 20 ms: many filter widths
        err = np.abs(y[edge:-edge] - want[edge:-edge]).max()
        assert err < 2e-3, (sr_in, sr_out, err)
        # a one-sample misalignment at the input rate would be this large: the bound above is far below it
        assert 2 * np.pi * f0 / sr_in > 0.2
        dc = O.sinc_resample(torch.ones(n), sr_in, sr_out).numpy()
        assert np.abs(dc[edge:-edge] - 1.0).max() < 2e-3
    x = np.random.default_rng(3).standard_normal(1000).astype(np.float32)
    assert np.array_equal(O.sinc_resample(_t(x), 24000, 24000).numpy(), x)
    t48 = np.arange(9600) / 48000.0
    sig = (np.sin(2 * np.pi * 440 * t48) + 0.5 * np.sin(2 * np.pi * 3100 * t48 + 1.0)).astype(np.float32)
    back = O.sinc_resample(O.sinc_resample(_t(sig), 48000, 24000), 24000, 48000).numpy()
    assert back.shape == sig.shape and np.abs(back[2000:-2000] - sig[2000:-2000]).max() < 5e-3


def test_badly_scaled_checkpoint_fixture(cfg, mc):
    """tests/golden/badscale*.npz (make_golden_badscale.py: THE REFERENCE on sopro_amd.weights.badly_scaled_*): the oracle reproduces
    the reference's refined tokens and waveform there too, and the fixture is as badly scaled as it says (residual-stream rows over
    more than five decades of RMS) - the GPU tests of the f16 range guard (tests/test_gpu_range.py) stand on it."""
    from sopro_amd.weights import badly_scaled_mimi, badly_scaled_sopro, synth_mimi_weights, synth_sopro_weights

    for name, overflow in (("badscale", False), ("badscale_overflow", True)):
        g = golden(name)
        seed = int(g["seed"])
        w = O.to_torch(badly_scaled_sopro(synth_sopro_weights(cfg, VOCAB, seed), cfg, overflow=overflow))
        toks = O.nar_refine(_t(g["cond"]), _t(g["rvq1"]), w, cfg)
        assert torch.equal(toks, _t(g["tokens"])), name
        assert float(g["stream_rms_min"]) < 2e-3 and float(g["stream_rms_max"]) > 5e2 and float(g["rel_margin"]) > 1e-6
        if not overflow:
            mw = O.to_torch(badly_scaled_mimi(synth_mimi_weights(mc, seed), mc))
            wav = O.decode_full(toks[0], mw, mc).reshape(-1)
            ref = _t(g["wav"])
            assert float((wav - ref).abs().max()) < 1e-5 * float(ref.abs().max())
