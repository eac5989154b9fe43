"""Range robustness of the split-precision kernels.  Every other parity fixture uses N(0, sigma) weights;
these use badly scaled checkpoints (sopro_amd.weights.badly_scaled_sopro / badly_scaled_mimi) whose expected results come from THE
REFERENCE (tests/golden/make_golden_badscale.py): residual-stream rows from 1e-3 to 7e+2 RMS within one refinement, `nar.pre` rows
over six decades, a codec decoder with layer gains of 16 / 1/16 and x 4 / x 1/4 transposed convolutions.

  * the f16 three-pass contractions must reproduce the reference's tokens there (per-row power-of-two staging scales in the
    fused-RMSNorm forms) with ZERO range events;
  * operands that do leave fp16's range must be COUNTED (sopro_gemm_split_ext.range_events), and the engine must repeat such a
    pass on the six-pass bf16 operands and still return the reference's tokens (the overflow fixture)."""
import numpy as np
import pytest
import torch

from conftest import VOCAB, golden

pytestmark = pytest.mark.gpu


def _t(a):
    return torch.from_numpy(np.asarray(a))


class _Tok:
    vocab_size = VOCAB

    def encode(self, text):
        return [1 + (ord(c) % 500) for c in text]


def _engine(cfg, mc, overflow):
    from sopro_amd import SoproTTS
    from sopro_amd.weights import badly_scaled_mimi, badly_scaled_sopro, synth_mimi_weights, synth_sopro_weights

    seed = int(golden("badscale")["seed"])
    wn = badly_scaled_sopro(synth_sopro_weights(cfg, VOCAB, seed), cfg, overflow=overflow)
    mn = badly_scaled_mimi(synth_mimi_weights(mc, seed), mc)
    return SoproTTS.from_weights(cfg, wn, mn, _Tok(), device="cuda:0"), wn, mn


def _audit(cfg, wn, cond, got, want, scale):
    """strict equality, or - for a decision whose fp32 margin is round-off sized against logits of ~1e3 - the oracle's own
    teacher-forced gap at every differing position must be below 1e-5 of the logit scale (the bar the other fixtures use, 1e-4, at
    their logit scale of ~10)"""
    if torch.equal(got, want):
        return 0
    from oracle import sopro_oracle as O

    n_off, gap = O.nar_audit(cond, got, O.to_torch(wn), cfg)
    assert gap < 1e-5 * scale, f"{n_off} refined tokens are off the oracle's arg-max by up to {gap:.3e} (logit scale {scale:.0f})"
    return n_off


def test_refinement_and_decode_on_a_badly_scaled_checkpoint_match_the_reference(cfg, mc):
    g = golden("badscale")
    tts, wn, _mn = _engine(cfg, mc, overflow=False)
    cond, rvq1, want = _t(g["cond"]), _t(g["rvq1"]), _t(g["tokens"])
    assert float(g["stream_rms_min"]) < 2e-3 and float(g["stream_rms_max"]) > 5e2  # the fixture is what it claims to be
    got = tts.model.nar_refine(cond, rvq1).cpu()
    assert getattr(tts.model, "range_fallbacks", 0) == 0, "the in-range fixture must not need the six-pass fallback"
    assert int(tts.model._recorded_blocks[("nar.range",)].values()[0]) == 0
    n_off = _audit(cfg, wn, cond, got, want, scale=1e3)
    assert n_off <= 2
    # waveform of the REFERENCE's tokens through the badly scaled decoder: 1e-4 of peak, like every other waveform fixture
    wav = tts.codec.decode_full(want[0].to("cuda:0")).reshape(-1).cpu()
    ref = _t(g["wav"])
    assert wav.shape == ref.shape
    err, peak = float((wav - ref).abs().max()), float(ref.abs().max())
    assert err < 1e-4 * peak, (err, peak)


def test_overflowing_operands_are_counted_and_the_pass_is_repeated_on_six_passes(cfg, mc):
    g = golden("badscale_overflow")
    tts, wn, _mn = _engine(cfg, mc, overflow=True)
    cond, rvq1, want = _t(g["cond"]), _t(g["rvq1"]), _t(g["tokens"])
    got = tts.model.nar_refine(cond, rvq1).cpu()
    assert getattr(tts.model, "range_fallbacks", 0) == 1, "a GELU output beyond fp16's range must trip the guard"
    _audit(cfg, wn, cond, got, want, scale=1e3)
    # ... and again from the recorded sequences (second call records, third replays): the guard word is part of the sequence
    for k in (2, 3):
        again = tts.model.nar_refine(cond, rvq1).cpu()
        assert torch.equal(again, got) and tts.model.range_fallbacks == k
    # the C stage API (what a host that is not this package writes): sopro_nar_refine_io leaves the events in the caller's word,
    # the host repeats the pass with safe = 1
    from sopro_amd.stages import StageEngine

    eng = StageEngine(tts)
    toks, events = eng.nar_refine_guarded(cond.to("cuda:0"), rvq1.to("cuda:0"))
    assert events > 0 and torch.equal(toks.cpu().long(), got)
    eng.close()


def test_the_scheduler_path_repeats_an_overflowing_pass(cfg, mc):
    """The whole-pass path of a scheduler: PipelinedSynthesizer queues refinement and decode back to back on the
    throughput partition and reads the guard word after the decoder's sync (SoproTTS.synthesize_batch); on the overflow checkpoint
    every pass must trip the guard, be repeated on the six-pass operands and decoded again - expected waveforms: the ORACLE's
    generate_tokens + decode_full on the same checkpoint (fp32 has the range).  Reference: src/sopro/model.py:307-347."""
    from oracle import sopro_oracle as O
    from sopro_amd.pipeline import PipelinedSynthesizer

    tts, wn, mn = _engine(cfg, mc, overflow=True)
    w, mw = O.to_torch(wn), O.to_torch(mn)
    rng = np.random.default_rng(17)
    ref_tq = torch.from_numpy(rng.integers(0, 2048, size=(20, 32)))
    ref, oref = tts.prepare_reference(ref_tokens_tq=ref_tq), O.prepare_reference(ref_tq, w, cfg)
    ids = [torch.from_numpy(rng.integers(1, VOCAB, size=n)) for n in (9, 14)]
    kw = dict(max_frames=11, top_p=0.0, temperature=1.0, anti_loop=False)
    want = []
    for x in ids:
        toks = O.generate_tokens(x, oref, w, cfg, style_strength=float(cfg.style_strength), **kw)
        want.append((toks, O.decode_full(toks, mw, mc)))
    job = dict(texts=[""] * 2, refs=[ref, ref], text_ids=ids, **kw)
    pipe = PipelinedSynthesizer(tts, lanes=2, ar_cus=64, ar_parts=1)
    try:
        outs = pipe.run([job] * 4)  # eager, recording and replayed passes on both lanes
        fallbacks = sum(getattr(l.model, "range_fallbacks", 0) for l in pipe.lanes)
    finally:
        pipe.close()
    assert fallbacks == 4, fallbacks  # every pass tripped the guard exactly once
    for out in outs:
        for b, (toks, owav) in enumerate(want):
            got = out[b].cpu().reshape(-1)
            assert got.numel() == owav.numel() == toks.shape[0] * 1920
            err, peak = float((got - owav.reshape(-1)).abs().max()), float(owav.abs().max())
            assert err < 1e-4 * peak, (b, err, peak)


def test_f16x3_range_word_at_the_operator_level():
    """sopro_gemm_f16x3 on its own: in-range operands leave the word alone; |a| > 8188 in a plain form, or an element 1000 x the
    row's first 32 in a fused-RMSNorm form, adds to it; rows of RMS 1e-4 ... 1e+4 keep the fused form's relative error at the
    1e-7 class (the per-row scale), where a constant scale loses four digits on the small rows."""
    from sopro_amd import hip

    dev = "cuda:0"
    g = torch.Generator(device="cpu").manual_seed(5)
    M, N, K = 256, 256, 384
    W = (torch.randn(N, K, generator=g) / K ** 0.5).to(dev)
    pw = hip.pack_w_f16x3(W)
    word = torch.zeros(4, dtype=torch.int32, device=dev)

    def run(A, rms):
        C = torch.empty(M, N, device=dev)
        hip.gemm(A, pw, C, M=M, N=N, K=K, rms_eps=1e-6 if rms else 0.0, range_events=word)
        torch.cuda.synchronize()
        return C

    A = torch.randn(M, K, generator=g).to(dev)
    scales = torch.logspace(-4, 4, M).to(dev)
    Ab = A * scales[:, None]
    Ad = Ab.double()
    want = (Ad * torch.rsqrt(Ad.pow(2).mean(-1, keepdim=True) + 1e-6)) @ W.double().T
    got = run(Ab, True)
    assert int(word[0]) == 0
    rel = float(((got.double() - want).abs().max(dim=1).values / want.abs().max(dim=1).values).max())
    assert rel < 2e-6, rel
    # plain form, in range / out of range
    got = run(A * 100.0, False)
    assert int(word[0]) == 0 and float((got.double() - (A.double() * 100.0) @ W.double().T).abs().max()) < 1e-3
    run(A * 1e4, False)
    n1 = int(word[0])
    assert n1 > 0
    # fused form: one element far beyond its row's leading 32
    Ac = A.clone()
    Ac[7, 200] = 1e6
    run(Ac, True)
    assert int(word[0]) > n1
