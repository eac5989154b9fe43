"""The stage-level C entry points (sopro_engine_*, sopro_ref_prepare, sopro_cond_prepare, sopro_ar_begin / _run_graph / _tokens,
sopro_nar_refine, sopro_mimi_decode: include/sopro_hip.h), driven through ctypes WITHOUT sopro_amd/model.py and
codec.py: a whole utterance from reference tokens + text ids to samples, against the reference's full-size fixture, the oracle and
the Python host (which issues the same kernels in the same order: results must be bit-identical)."""
import numpy as np
import pytest
import torch

from conftest import golden
from oracle import sopro_oracle as O
from sopro_amd.stages import StageEngine

pytestmark = pytest.mark.gpu
GREEDY = dict(top_p=0.0, temperature=1.0, anti_loop=False)


def _t(a):
    return torch.from_numpy(np.asarray(a))


@pytest.fixture(scope="module")
def eng(tts_noeos):
    e = StageEngine(tts_noeos)
    yield e
    e.close()


def test_whole_utterance_through_the_c_stages_matches_the_reference(eng, tts_noeos, cfg, w_noeos):
    g = golden("full200")
    tts = tts_noeos
    maxf = int(g["max_frames"])
    want = _t(g["tokens"].astype(np.int64))
    ref = tts.prepare_reference(ref_tokens_tq=_t(g["ref_tq"]))
    # reference preparation and conditioning through the library's sequences alone (tokens + text ids in) ...
    voice = eng.reference(_t(g["ref_tq"]))
    prep = eng.conditioning(_t(g["ids"]), voice, maxf, style_strength=1.0)
    # ... which the Python host marshals to as well: identical bits
    pprep = tts.model.prepare_conditioning(_t(g["ids"]), ref, max_frames=maxf, style_strength=1.0)
    assert torch.equal(voice["sv"], ref.sv_ref) and torch.equal(voice["ref_seq"], ref.ref_seq[0])
    assert torch.equal(prep["cond_ar"], pprep["cond_ar"]) and torch.equal(prep["txt_seq"], pprep["txt_seq"])
    hist, feos = eng.ar_generate(prep["cond_ar"], prep["txt_seq"], None, **GREEDY)
    assert int(feos[0]) == -1 and hist.shape == (1, maxf + 1)
    assert torch.equal(hist[0].cpu().long(), want[:, 0]), "codebook-0 tokens differ from the reference"
    toks = eng.nar_refine(prep["cond_ar"], hist)
    assert toks.shape == (1, maxf + 1, 32) and torch.equal(toks[0, :, 0].cpu().long(), want[:, 0])
    assert torch.equal(toks[0].cpu().long(), want), "refined tokens differ from the reference's fixture (strict: no audit)"
    wav = eng.mimi_decode(_t(g["tokens"].astype(np.int64)).unsqueeze(0).to(eng.device))
    assert tuple(wav.shape) == (1, (maxf + 1) * 1920)
    assert float((wav[0].cpu() - _t(g["wav"])).abs().max()) < 1e-4 * float(np.abs(g["wav"]).max())
    # the Python host issues the same launches: identical bits
    ptoks = tts.model.generate_tokens(_t(g["ids"]), ref, max_frames=maxf, style_strength=1.0, **GREEDY)
    assert torch.equal(ptoks.cpu(), toks[0].cpu().long())
    # (the Python host runs few-row contractions split-K - another summation order -, so samples agree to round-off, not bit for bit)
    pwav = tts.codec.decode_full(want)
    assert float((pwav.reshape(-1) - wav.reshape(-1)).abs().max()) < 2e-5 * float(np.abs(g["wav"]).max())


def test_c_stages_on_a_ragged_sampled_batch(eng, tts_noeos):
    """B = 5 rows, ragged text lengths, stochastic decoding with a pinned (seed, nonce), a second generation on the same
    engine (recorded frame graph replayed on a fresh state), NAR with per-row lengths: equal to the Python host."""
    tts = tts_noeos
    rng = np.random.default_rng(91)
    ids = [torch.from_numpy(rng.integers(0, 512, size=n)) for n in (33, 64, 17, 50, 41)]
    ref = tts.prepare_reference(ref_tokens_tq=torch.from_numpy(rng.integers(0, 2048, size=(60, 32))))
    prep = tts.model.prepare_conditioning_batch(ids, [ref] * 5, max_frames=39, style_strength=1.0)
    kw = dict(top_p=0.9, temperature=1.05, anti_loop=True)
    for seed in (3, 4):
        hist, _ = eng.ar_generate(prep["cond_ar"], prep["txt_seq"], prep["text_lens"], seed=tts.model.seed, nonce=seed, **kw)
        ph, _ = tts.model.ar_generate_batch(prep["cond_ar"], prep["txt_seq"], prep["text_lens"], max_frames=39, seed=seed, **kw)
        assert torch.equal(hist[:, : ph.shape[1]].cpu(), ph.cpu()), seed
    lens = torch.tensor([40, 25, 33, 40, 8], dtype=torch.int32, device=eng.device)
    toks = eng.nar_refine(prep["cond_ar"], hist.clamp(max=2047), lens)
    ptoks = tts.model.nar_refine(prep["cond_ar"][:, :40], hist[:, :40].clamp(max=2047), lens=lens.tolist())
    for b, n in enumerate(lens.tolist()):
        assert torch.equal(toks[b, :n].cpu().long(), ptoks[b, :n].cpu()), b
    wav = eng.mimi_decode(ptoks)
    pw = tts.codec.decode_batch(ptoks)
    assert float((wav - pw).abs().max()) < 2e-5 * float(pw.abs().max())


def test_c_stages_on_a_32x200_batch(eng, tts_noeos, cfg, w_noeos):
    """BASELINE configs[1] through the C stage API alone: 32 utterances x 200 frames, 4 voices, greedy.  Every row's tokens
    equal the Python host's (same kernels, same order); rows 2 and 29 are checked against their own oracle runs."""
    tts = tts_noeos
    torch.set_num_threads(8)
    rng = np.random.default_rng(93)
    B, maxf = 32, 199
    ids = [torch.from_numpy(rng.integers(0, 512, size=64)) for _ in range(B)]
    refs_tq = [torch.from_numpy(rng.integers(0, 2048, size=(150, 32))) for _ in range(4)]
    voices = [tts.prepare_reference(ref_tokens_tq=r) for r in refs_tq]
    refs = [voices[b % 4] for b in range(B)]
    prep = tts.model.prepare_conditioning_batch(ids, refs, max_frames=maxf, style_strength=1.0)
    hist, feos = eng.ar_generate(prep["cond_ar"], prep["txt_seq"], prep["text_lens"], **GREEDY)
    assert hist.shape == (B, maxf + 1) and bool((feos < 0).all())
    toks = eng.nar_refine(prep["cond_ar"], hist)
    ptoks = tts.model.generate_tokens_batch(ids, refs, max_frames=maxf, style_strength=1.0, **GREEDY)
    for b in range(B):
        assert torch.equal(toks[b].cpu().long(), ptoks[b].cpu()), b
    wav = eng.mimi_decode(toks)
    pw = tts.codec.decode_batch(torch.stack(ptoks))
    assert tuple(wav.shape) == (B, (maxf + 1) * 1920)
    assert float((wav - pw).abs().max()) < 2e-5 * float(pw.abs().max())
    for b in (2, 29):
        oref = O.prepare_reference(refs_tq[b % 4], w_noeos, cfg)
        want = O.generate_tokens(ids[b], oref, w_noeos, cfg, max_frames=maxf, style_strength=1.0, **GREEDY)
        got = toks[b].cpu().long()
        assert torch.equal(got[:, 0], want[:, 0]), b
        if not torch.equal(got, want):
            oprep = O.prepare_conditioning(ids[b], oref, w_noeos, cfg, max_frames=maxf, style_strength=1.0)
            n_off, gap = O.nar_audit(oprep["cond_ar"][:, : maxf + 1], got.unsqueeze(0), w_noeos, cfg)
            assert gap < 1e-4, (b, n_off, gap)


@pytest.mark.parametrize("trim", ["none", "legacy"])
def test_streaming_decode_through_the_c_stage(eng, tts_noeos, trim):
    """sopro_mimi_decode_stream (+ _trim for the legacy policy) against the Python host's MimiStreamDecoder over 40 chunks of
    6 frames: the cache passes the 250-position window ("none": eviction to the last 249) or keeps growing ("legacy")."""
    from sopro_amd.codec import MimiStreamDecoder
    from sopro_amd.stages import StageStreamDecoder

    g = golden("full400")
    codes = _t(g["tokens"].astype(np.int64))[:240]
    cdec = StageStreamDecoder(eng, trim=trim)
    pdec, pst = MimiStreamDecoder(tts_noeos.codec, trim=trim), None
    worst, scale = 0.0, float(np.abs(g["wav"]).max())
    for i in range(0, 240, 6):
        cw = cdec.decode_step(codes[i:i + 6])
        pw, pst = pdec.decode_step(codes[i:i + 6], pst)
        assert cw.shape == pw.shape == (1, 6 * 1920)
        worst = max(worst, float((cw - pw).abs().max()))
    assert worst < 2e-5 * scale, worst  # few-row contractions: the Python host splits K, the C sequence does not (round-off only)
    assert cdec.st.pos == pst.pos and cdec.st.kv_len == pst.kv_len and cdec.st.evict == int(pst.evict)


def test_library_profiler_times_the_stage_launches(tts_noeos):
    """sopro_prof_enable / sopro_prof_collect (csrc/prof.hip): while bench.py's profiler is attached the launches of the NAR
    and Mimi sequences are bracketed inside the library and come back per kernel family with their algorithmic flops."""
    from sopro_amd import hip

    tts = tts_noeos
    rng = np.random.default_rng(5)
    toks = torch.from_numpy(rng.integers(0, 2048, size=(2, 40, 32))).to(tts.device)
    want = tts.codec.decode_batch(toks).clone()
    cond = torch.randn(2, 40, 384, device=tts.device)
    p = hip.Profiler()
    hip.set_profiler(p)
    try:
        got = tts.codec.decode_batch(toks).clone()
        tts.model.nar_refine(cond, toks[:, :, 0].contiguous())
    finally:
        hip.set_profiler(None)
    fam = p.summary()
    assert torch.equal(got, want)  # the timed (eager) sequence is the recorded one
    for k in ("gemm_bf16x3_kernel", "gemm_f16x3_kernel", "attention_split_kernel", "seanet_tail_kernel", "seanet_res128_kernel", "seanet_up128_kernel"):
        assert k in fam and fam[k]["launches"] > 0 and fam[k]["flops"] > 0 and fam[k]["ms_all"] > 0, (k, fam.get(k))
    assert fam["seanet_tail_kernel"]["launches"] == 1 and fam["seanet_tail_kernel"]["flops"] == 2.0 * 2 * 40 * 1920 * (3 * 64 * 32 + 32 * 64 + 3 * 64)
    assert hip.Profiler().summary() == {}  # collected records are gone


def test_missing_tensor_is_reported_by_name(tts_noeos):
    import ctypes as C
    from sopro_amd import hip
    from sopro_amd.stages import engine_cfg

    lib = hip.load()
    m, codec = tts_noeos.model, tts_noeos.codec
    cfg = engine_cfg(m.cfg, codec.mc, m.gates, [(1.0, 0.0)] * 4, [0.0] * 32, 0.0, 1024)
    h = C.c_void_p()
    assert lib.sopro_engine_create(C.byref(cfg), C.byref(h)) == 0
    try:
        assert lib.sopro_engine_finalize(h, None) == -2
        assert b"ar.blocks.0.glu.w" in lib.sopro_last_error()
        assert lib.sopro_ar_run_graph(h, 1, None) == -2
    finally:
        lib.sopro_engine_destroy(h)


def test_engine_built_by_the_c_loader_from_reference_state_dict_files(tmp_path, cfg, mc, sopro_np_noeos, mimi_np):
    """Round 4 ("names = reference state_dict keys"): the library reads model.safetensors (SoproTTSModel.state_dict() keys,
    config JSON in the header: src/sopro/hub.py:30-52) and the Mimi codec's safetensors (HuggingFace keys) itself, repacks on the
    host and builds the engine - no Python packing.  The reference's 200-frame fixture through that engine: all 32 codebooks
    strict-equal, waveform within 1e-4 of peak."""
    from safetensors.numpy import save_file

    from sopro_amd.stages import CheckpointEngine
    from sopro_amd.weights import save_sopro_checkpoint

    sp, mp = str(tmp_path / "model.safetensors"), str(tmp_path / "mimi.safetensors")
    save_sopro_checkpoint(sp, sopro_np_noeos, cfg)
    save_file({k: np.require(v, requirements="C") for k, v in mimi_np.items()}, mp)
    e = CheckpointEngine(sp, mp)
    try:
        g = golden("full200")
        maxf = int(g["max_frames"])
        want = _t(g["tokens"].astype(np.int64))
        voice = e.reference(_t(g["ref_tq"]))
        prep = e.conditioning(_t(g["ids"]), voice, maxf, style_strength=1.0)
        hist, feos = e.ar_generate(prep["cond_ar"], prep["txt_seq"], None, **GREEDY)
        assert int(feos[0]) == -1 and torch.equal(hist[0].cpu().long(), want[:, 0]), "codebook-0 tokens differ from the reference"
        toks = e.nar_refine(prep["cond_ar"], hist)
        assert torch.equal(toks[0].cpu().long(), want), "refined tokens differ from the reference's fixture"
        wav = e.mimi_decode(toks)
        assert float((wav[0].cpu() - _t(g["wav"])).abs().max()) < 1e-4 * float(np.abs(g["wav"]).max())
    finally:
        e.close()
