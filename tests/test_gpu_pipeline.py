"""Stage-level and end-to-end parity of the MI355X engine (MI355X only), through the public
SoproTTS / SoproTTSModel / MimiCodec mirrors:
  * against the golden fixtures produced by THE REFERENCE (tests/golden/*.npz), and
  * against the CPU oracle on further seeded inputs (batched / ragged cases the reference cannot run),
  * plus size-independent properties at the benchmark's full size (32 x 200 frames).
Bars: token ids bit-exact under greedy decode; logits atol 2e-4; waveform atol 1e-4 * max|wav| (fp32)."""
import numpy as np
import pytest
import torch

from conftest import FakeTok, golden
from oracle import sopro_oracle as O

pytestmark = pytest.mark.gpu


def _t(a):
    return torch.from_numpy(np.asarray(a))


def _err(a, b):
    return float((a.detach().float().cpu() - b.detach().float().cpu()).abs().max())


@pytest.fixture(scope="module")
def ref_prep(tts):
    g = golden("prep")
    ref = tts.prepare_reference(ref_tokens_tq=_t(g["ref_tq"]))
    prep = tts.model.prepare_conditioning(_t(g["ids"]), ref, max_frames=int(g["max_frames"]), style_strength=float(g["style_strength"]))
    return g, ref, prep


def test_prepare_reference_matches_reference(ref_prep):
    g, ref, _ = ref_prep
    assert tuple(ref.ref_tokens_btq.shape) == (1, 30, 32) and ref.ref_kv_caches[0]["key_padding_mask"] is None
    assert tuple(ref.ref_kv_caches[0]["k"].shape) == (1, 2, 30, 192)
    assert _err(ref.sv_ref, _t(g["sv_ref"])) < 5e-6
    assert _err(ref.ref_seq, _t(g["ref_seq"])) < 5e-5
    assert _err(ref.ref_kv_caches[0]["k"], _t(g["ref_k0"])) < 5e-5
    assert _err(ref.ref_kv_caches[2]["v"], _t(g["ref_v2"])) < 5e-5


def test_encode_speaker_matches_reference(tts):
    """SoproTTS.encode_speaker against the reference facade's own method (src/sopro/model.py:457-475;
    tests/golden/make_golden_speaker.py) under its three crop policies, and the error behaviour of encode_reference."""
    g = golden("speaker")
    ref_tq = _t(g["ref_tq"])
    for name, secs in (("default", None), ("sec4", 4.0), ("nocrop", 0.0)):
        sv = tts.encode_speaker(ref_tokens_tq=ref_tq, ref_seconds=secs)
        assert tuple(sv.shape) == (192,) and sv.device.type == "cuda"
        assert _err(sv, _t(g["sv_" + name])) < 5e-6, name
    # the vector prepare_reference stores for the same voice
    assert _err(tts.prepare_reference(ref_tokens_tq=ref_tq).sv_ref.squeeze(0), _t(g["sv_default"])) < 5e-6
    with pytest.raises(RuntimeError):
        tts.encode_speaker()


def test_prepare_conditioning_matches_reference(ref_prep):
    g, _, prep = ref_prep
    assert _err(prep["txt_seq"], _t(g["txt_seq"])) < 5e-5
    assert _err(prep["txt_pool"], _t(g["txt_pool"])) < 5e-5
    assert _err(prep["cond_ar"], _t(g["cond_ar"])) < 1e-4
    assert prep["text_mask"].dtype == torch.bool and tuple(prep["text_mask"].shape) == (1, 19)


def test_prepare_conditioning_batch_ragged_matches_oracle(tts, cfg, w):
    rng = np.random.default_rng(5)
    ids = [torch.from_numpy(rng.integers(0, 512, size=n)) for n in (9, 23, 14)]
    refs_tq = [torch.from_numpy(rng.integers(0, 2048, size=(n, 32))) for n in (20, 33, 27)]
    refs = [tts.prepare_reference(ref_tokens_tq=r, ref_seconds=0) for r in refs_tq]
    out = tts.model.prepare_conditioning_batch(ids, refs, max_frames=30, style_strength=1.0)
    for b in range(3):
        oref = O.prepare_reference(refs_tq[b], w, cfg)
        op = O.prepare_conditioning(ids[b], oref, w, cfg, max_frames=30, style_strength=1.0)
        assert _err(out["cond_ar"][b], op["cond_ar"][0]) < 1e-4, b
        assert _err(out["txt_seq"][b, : ids[b].numel()], op["txt_seq"][0]) < 5e-5, b


def test_prepare_conditioning_batch_shared_voices(tts):
    """Rows that pass the same PreparedReference share its cached K / V (zero batch stride for one voice, one gather for a
    mix): same conditioning as with one private copy per row."""
    rng = np.random.default_rng(6)
    ids = [torch.from_numpy(rng.integers(0, 512, size=n)) for n in (9, 23, 14, 17)]
    ra = tts.prepare_reference(ref_tokens_tq=torch.from_numpy(rng.integers(0, 2048, size=(20, 32))), ref_seconds=0)
    rb = tts.prepare_reference(ref_tokens_tq=torch.from_numpy(rng.integers(0, 2048, size=(31, 32))), ref_seconds=0)
    for refs in ([ra, ra, ra, ra], [ra, rb, ra, rb], [rb, rb, ra, rb]):
        out = tts.model.prepare_conditioning_batch(ids, refs, max_frames=30, style_strength=1.0)
        for b in range(4):
            one = tts.model.prepare_conditioning_batch([ids[b]], [refs[b]], max_frames=30, style_strength=1.0)
            assert _err(out["cond_ar"][b], one["cond_ar"][0]) < 1e-5, (b, [id(r) for r in refs])


@pytest.mark.parametrize("use_graph", [False, True])
def test_ar_step_teacher_forced_logits_match_reference(tts, use_graph):
    """ARRVQ1Generator.step for B=2 with a ragged text mask (src/sopro/nn/generator.py:98-130)."""
    from sopro_amd.model import _ARRun

    g = golden("ar_teacher")
    x, txt, mask = _t(g["x"]), _t(g["txt"]), _t(g["mask"])
    B, T, D = x.shape
    m = tts.model
    old = m.use_graph
    m.use_graph = use_graph
    m._ar_cache.clear()
    try:
        run = _ARRun(m, torch.zeros(B, T, D), txt, mask.sum(1).to(torch.int32), top_p=0.0, temperature=1.0, anti_loop=False,
                     min_gen_frames=10 ** 6)
        xd = x.to(m.device)
        worst = 0.0
        for t in range(T):
            with torch.cuda.stream(m.stream):
                run.plan.x[0].copy_(xd[:, t])
            run.advance(1)
            m.stream.synchronize()
            worst = max(worst, _err(run.plan.logits, _t(g["logits"][:, t])))
        assert worst < 2e-4, worst
    finally:
        m.use_graph = old
        m._ar_cache.clear()


def test_ar_greedy_tokens_bit_exact(tts, ref_prep):
    g, _, prep = ref_prep
    gg = golden("ar_greedy")
    toks = [tk for _t2, tk, _e in tts.model.ar_stream(prep, max_frames=int(g["max_frames"]), top_p=0.0, temperature=1.0, anti_loop=False)]
    assert toks == gg["tokens"].tolist()
    toks6 = [tk for _t2, tk, _e in tts.model.ar_stream(prep, max_frames=int(g["max_frames"]), top_p=0.0, temperature=1.0,
                                                        anti_loop=False, lookahead=6)]
    assert toks6 == toks


def test_ar_batch_rows_are_independent_and_match_oracle(tts, cfg, w):
    rng = np.random.default_rng(11)
    ids = [torch.from_numpy(rng.integers(0, 512, size=n)) for n in (17, 8, 30)]
    refs_tq = [torch.from_numpy(rng.integers(0, 2048, size=(25, 32))) for _ in range(3)]
    refs = [tts.prepare_reference(ref_tokens_tq=r) for r in refs_tq]
    prep = tts.model.prepare_conditioning_batch(ids, refs, max_frames=24, style_strength=1.0)
    hist, lens = tts.model.ar_generate_batch(prep["cond_ar"], prep["txt_seq"], prep["text_lens"], max_frames=24, top_p=0.0,
                                             temperature=1.0, anti_loop=False)
    for b in range(3):
        oref = O.prepare_reference(refs_tq[b], w, cfg)
        op = O.prepare_conditioning(ids[b], oref, w, cfg, max_frames=24, style_strength=1.0)
        want = [tk for _a, tk, _e in O.ar_generate(op, w, cfg, max_frames=24, top_p=0.0, temperature=1.0, anti_loop=False)]
        got = hist[b, : len(want)].cpu().tolist()
        assert got == want, b


def test_eos_rule_and_generate_tokens(cfg, sopro_np, mimi_np):
    """EOS sampled before/after min_gen_frames (src/sopro/model.py:301-305, 385-390)."""
    from sopro_amd import SoproTTS

    ge, g = golden("ar_eos"), golden("prep")
    wts = dict(sopro_np)
    hb = sopro_np["ar.head.bias"].copy()
    hb[2048] = float(ge["eos_bias"])
    wts["ar.head.bias"] = hb
    t2 = SoproTTS.from_weights(cfg, wts, mimi_np, FakeTok(), device="cuda:0")
    ref = t2.prepare_reference(ref_tokens_tq=_t(g["ref_tq"]))
    prep = t2.model.prepare_conditioning(_t(g["ids"]), ref, max_frames=int(g["max_frames"]), style_strength=float(g["style_strength"]))
    kw = dict(max_frames=int(g["max_frames"]), top_p=0.0, temperature=float(ge["temperature"]), anti_loop=False,
              min_gen_frames=int(ge["min_gen_frames"]))
    ev = list(t2.model.ar_stream(prep, **kw))
    assert [tk for _a, tk, _e in ev] == ge["tokens"].tolist()
    toks = t2.model.generate_tokens(_t(g["ids"]), ref, style_strength=float(g["style_strength"]), **kw)
    assert torch.equal(toks.cpu(), _t(ge["gen_tokens"]))


def test_batch_with_rows_stopping_at_different_frames(cfg, sopro_np, mimi_np, w):
    """Ragged EOS in a batch: every row must equal its own single-utterance reference run
    (generate_tokens cut rule, src/sopro/model.py:385-390), rows keep stepping after they stopped."""
    from sopro_amd import SoproTTS

    wts = dict(sopro_np)
    hb = sopro_np["ar.head.bias"].copy()
    hb[2048] = 3.9
    wts["ar.head.bias"] = hb
    t2 = SoproTTS.from_weights(cfg, wts, mimi_np, FakeTok(), device="cuda:0")
    w2 = dict(w)
    w2["ar.head.bias"] = torch.from_numpy(hb)
    rng = np.random.default_rng(61)
    ids = [torch.from_numpy(rng.integers(0, 512, size=n)) for n in (19, 11, 26, 7)]
    refs_tq = [torch.from_numpy(rng.integers(0, 2048, size=(22, 32))) for _ in ids]
    refs = [t2.prepare_reference(ref_tokens_tq=r) for r in refs_tq]
    kw = dict(max_frames=40, top_p=0.0, temperature=0.8, anti_loop=False, min_gen_frames=6)
    got = t2.model.generate_tokens_batch(ids, refs, style_strength=1.0, **kw)
    lens = []
    for b in range(len(ids)):
        oref = O.prepare_reference(refs_tq[b], w2, cfg)
        want = O.generate_tokens(ids[b], oref, w2, cfg, style_strength=1.0, **kw)
        lens.append(int(want.shape[0]))
        assert tuple(got[b].shape) == tuple(want.shape), (b, tuple(got[b].shape), tuple(want.shape))
        assert torch.equal(got[b][:, 0].cpu(), want[:, 0]), b
        if not torch.equal(got[b].cpu(), want):  # a refined token may differ only where the oracle itself sees a near-tie
            op = O.prepare_conditioning(ids[b], oref, w2, cfg, max_frames=40, style_strength=1.0)
            n_off, gap = O.nar_audit(op["cond_ar"][:, : want.shape[0]], got[b].cpu().unsqueeze(0), w2, cfg)
            assert gap < 1e-4, (b, n_off, gap)
    assert len(set(lens)) > 1, f"fixture is not ragged: {lens}"


def test_nar_refine_tokens_bit_exact(tts, ref_prep):
    g, _, prep = ref_prep
    gn = golden("nar")
    T = int(gn["T"])
    toks = tts.model.nar_refine(prep["cond_ar"][:, :T], _t(gn["rvq1"]))
    assert toks.dtype == torch.long and tuple(toks.shape) == (1, T, 32)
    mism = int((toks.cpu() != _t(gn["tokens"])).sum())
    assert mism == 0, f"{mism} of {T * 31} refined tokens differ (min reference margin {float(gn['min_margin']):.2e})"


def test_nar_refine_ragged_batch_matches_oracle(tts, cfg, w, ref_prep):
    _, _, prep = ref_prep
    rng = np.random.default_rng(21)
    lens = [29, 12, 40]
    cond = prep["cond_ar"][:, :40].repeat(3, 1, 1) * torch.tensor([1.0, 0.9, 1.1], device=prep["cond_ar"].device)[:, None, None]
    rvq1 = torch.from_numpy(rng.integers(0, 2048, size=(3, 40)))
    toks = tts.model.nar_refine(cond, rvq1, lens=lens)
    for b, n in enumerate(lens):
        want = O.nar_refine(cond[b: b + 1, :n].cpu(), rvq1[b: b + 1, :n], w, cfg)
        if not torch.equal(toks[b, :n].cpu(), want[0]):  # only audited near-ties may differ (no blanket mismatch budget)
            n_off, gap = O.nar_audit(cond[b: b + 1, :n].cpu(), toks[b: b + 1, :n].cpu(), w, cfg)
            assert gap < 1e-4, (b, n_off, gap)


def test_mimi_decode_matches_hf_reference(tts):
    g = golden("mimi")
    scale = float(np.abs(g["wav32"]).max())
    w8 = tts.codec.decode_full(_t(g["tok8"]))
    assert tuple(w8.shape) == (1, 1, 8 * 1920)
    assert _err(w8, _t(g["wav8"])) < 1e-4 * scale
    w32 = tts.codec.decode_full(_t(g["tok32"]))
    assert _err(w32, _t(g["wav32"])) < 1e-4 * scale
    # batch of two (ragged: the shorter one is a causal prefix problem)
    codes = torch.zeros(2, 32, 32, dtype=torch.long)
    codes[0] = _t(g["tok32"])
    codes[1, :8] = _t(g["tok8"])
    wb = tts.codec.decode_batch(codes)
    assert _err(wb[0], _t(g["wav32"]).reshape(-1)) < 1e-4 * scale
    assert _err(wb[1, : 8 * 1920], _t(g["wav8"]).reshape(-1)) < 1e-4 * scale


def test_synthesize_and_stream_match_reference(tts, ref_prep):
    g, ref, _ = ref_prep
    ge = golden("e2e")
    tts.tokenizer.table["hello"] = g["ids"].tolist()
    kw = dict(max_frames=int(ge["max_frames"]), top_p=0.0, temperature=1.0, anti_loop=False, style_strength=float(g["style_strength"]))
    wav = tts.synthesize("hello", ref=ref, **kw)
    want = _t(ge["wav"])
    scale = float(want.abs().max())
    assert tuple(wav.shape) == tuple(want.shape)
    assert _err(wav, want) < 1e-4 * scale
    chunks = list(tts.stream("hello", ref=ref, chunk_frames=6, **kw))
    assert [int(c.shape[1]) for c in chunks] == ge["chunk_sizes"].tolist() and chunks[0].dim() == 2
    assert _err(torch.cat(chunks, dim=1), _t(ge["stream"])) < 1e-4 * scale
    with pytest.raises(RuntimeError):
        tts.synthesize("hello")
    with pytest.raises(RuntimeError):
        tts.synthesize("hello", ref_tokens_tq=_t(g["ref_tq"]), ref_audio_path="x.wav")


def test_synthesize_batch_equals_single_calls(tts, ref_prep):
    g, ref, _ = ref_prep
    rng = np.random.default_rng(31)
    ids = [torch.from_numpy(rng.integers(0, 512, size=n)) for n in (12, 19, 7)]
    kw = dict(max_frames=16, top_p=0.0, temperature=1.0, anti_loop=False, style_strength=1.0)
    for i, x in enumerate(ids):
        tts.tokenizer.table[f"u{i}"] = x.tolist()
    singles = [tts.synthesize(f"u{i}", ref=ref, **kw) for i in range(3)]
    batch = tts.synthesize_batch([f"u{i}" for i in range(3)], [ref] * 3, **kw)
    for a, b in zip(singles, batch):
        assert tuple(a.shape) == tuple(b.shape)
        assert _err(a, b) < 1e-4 * float(a.abs().max())


def test_two_lane_pipeline_equals_sequential(tts, ref_prep):
    """CU-partitioned two-lane pipelining returns exactly what the sequential path returns, in job order."""
    from sopro_amd.pipeline import PipelinedSynthesizer

    _, ref, _ = ref_prep
    rng = np.random.default_rng(51)
    jobs = []
    for j in range(3):
        ids = [torch.from_numpy(rng.integers(0, 512, size=n)) for n in (9 + j, 15, 6)]
        jobs.append(dict(texts=[""] * 3, refs=[ref] * 3, text_ids=ids, max_frames=14, top_p=0.0, temperature=1.0, anti_loop=False,
                         style_strength=1.0))
    seq = [tts.synthesize_batch(**j) for j in jobs]
    pipe = PipelinedSynthesizer(tts, lanes=2, ar_cus=64)
    try:
        par = pipe.run(jobs)
    finally:
        pipe.close()
    for a, b in zip(seq, par):
        for x, y in zip(a, b):
            assert tuple(x.shape) == tuple(y.shape)
            assert torch.equal(x, y), "pipelined result differs from the sequential path"


def test_coalesced_passes_equal_the_jobs_one_by_one(tts, ref_prep):
    """Round 3: the pipeline generates / refines / decodes pairs (triples) of compatible jobs as one pass.
    (a) STOCHASTIC decoding with per-job seeds at a size where no contraction changes its K-split with the row count (32
    utterances x 40 frames, S = 64): every utterance gets, bit for bit, the audio it gets when its job runs alone - each row keeps
    its own job's nonce and its index within that job in the sampler's counter (sopro_ar_state.row_id).
    (b) Small ragged jobs of different sizes under greedy decoding, an incompatible job (other top_p) in the middle that must stay
    on its own, an odd job count: grouping as specified, results within the batch-vs-single tolerance (few-row contractions run
    split-K by row count, i.e. another summation order)."""
    from sopro_amd.pipeline import PipelinedSynthesizer

    _, ref, _ = ref_prep
    rng = np.random.default_rng(53)
    big = []
    for j in range(3):
        ids = [torch.from_numpy(rng.integers(0, 512, size=64)) for _ in range(32)]
        big.append(dict(texts=[""] * 32, refs=[ref] * 32, text_ids=ids, max_frames=39, top_p=0.9, temperature=1.05, anti_loop=True,
                        style_strength=1.0, seed=100 + j))
    small = []
    for j in range(7):
        n = 3 if j % 2 == 0 else 5
        ids = [torch.from_numpy(rng.integers(0, 512, size=int(k))) for k in rng.integers(5, 20, size=n)]
        small.append(dict(texts=[""] * n, refs=[ref] * n, text_ids=ids, max_frames=16, top_p=0.8 if j == 3 else 0.0, temperature=1.0, anti_loop=False,
                          style_strength=1.0, seed=7))
    seq_big = [tts.synthesize_batch(**j) for j in big]
    seq_small = [tts.synthesize_batch(**j) for j in small]
    assert not torch.equal(seq_big[0][0], seq_big[1][0])  # the draws differ between jobs (and between rows)
    pipe = PipelinedSynthesizer(tts, lanes=2, ar_cus=64)
    try:
        for co in (2, 3):
            par = pipe.run(big, coalesce=co)
            for a, b in zip(seq_big, par):
                assert len(a) == len(b) == 32
                for x, y in zip(a, b):
                    assert tuple(x.shape) == tuple(y.shape) and torch.equal(x, y), f"coalesce={co}: a coalesced utterance differs from its own job's result"
        assert [g for g, _m, _s in pipe._coalesce(small, 2)] == [[0, 1], [2], [3], [4, 5], [6]]
        par = pipe.run(small, coalesce=2)
        for a, b in zip(seq_small[:3] + seq_small[4:], par[:3] + par[4:]):  # (job 3 samples: its own pass, compared above in kind)
            assert len(a) == len(b)
            for x, y in zip(a, b):
                assert tuple(x.shape) == tuple(y.shape) and _err(x, y) < 1e-4 * float(x.abs().max())
        for x, y in zip(seq_small[3], par[3]):
            assert torch.equal(x, y)
    finally:
        pipe.close()


def test_four_lanes_two_generation_slots_fill_streams_and_order(tts, ref_prep):
    """The bench configuration in small (4 lanes, two AR phases at a time on one shared partition): results equal the sequential
    path in job order, run after run, and the pipeline-fill streams (a share of the whole chip for the FIRST phase of each
    generation slot) go to at most one phase per slot and run - whichever lane takes the slot first, not job 0 / 1."""
    from sopro_amd.pipeline import PipelinedSynthesizer

    _, ref, _ = ref_prep
    rng = np.random.default_rng(52)
    jobs = []
    for j in range(7):
        ids = [torch.from_numpy(rng.integers(0, 512, size=n)) for n in (7 + j, 12, 5, 9)]
        jobs.append(dict(texts=[""] * 4, refs=[ref] * 4, text_ids=ids, max_frames=10 + (j % 3), top_p=0.0, temperature=1.0, anti_loop=False,
                         style_strength=1.0))
    seq = [tts.synthesize_batch(**j) for j in jobs]
    pipe = PipelinedSynthesizer(tts, lanes=4, ar_cus=64, ar_parts=2, ar_shared=True)
    try:
        for _ in range(3):
            par = pipe.run(jobs)
            assert 1 <= len(pipe.fill_jobs) <= 2, pipe.fill_jobs
            for a, b in zip(seq, par):
                for x, y in zip(a, b):
                    assert torch.equal(x, y), "pipelined result differs from the sequential path"
    finally:
        pipe.close()
    again = [tts.synthesize_batch(**j) for j in jobs[:2]]  # the caller's engine got its own streams back
    for a, b in zip(seq, again):
        for x, y in zip(a, b):
            assert torch.equal(x, y)


def test_full_size_properties_32x200(tts, cfg, sopro_np, mimi_np):
    """BASELINE config 2 size: determinism, batch invariance, causal-prefix property of the decoder."""
    from sopro_amd import SoproTTS
    from sopro_amd.weights import synth_sopro_weights

    big = SoproTTS.from_weights(cfg, synth_sopro_weights(cfg, 512, 1234, suppress_eos=True), mimi_np, FakeTok(), device="cuda:0")
    rng = np.random.default_rng(41)
    B, F = 32, 199
    ids = [torch.from_numpy(rng.integers(0, 512, size=64)) for _ in range(B)]
    ref = big.prepare_reference(ref_tokens_tq=torch.from_numpy(rng.integers(0, 2048, size=(150, 32))))
    kw = dict(max_frames=F, top_p=0.0, temperature=1.0, anti_loop=False)
    t1 = big.model.generate_tokens_batch(ids, [ref] * B, **kw)
    t2 = big.model.generate_tokens_batch(ids, [ref] * B, **kw)
    assert all(int(x.shape[0]) == F + 1 for x in t1)
    assert all(torch.equal(a, b) for a, b in zip(t1, t2)), "greedy generation is not deterministic"
    solo = big.model.generate_tokens(ids[5], ref, **kw)
    assert torch.equal(solo, t1[5]), "row 5 of the batch differs from a batch-of-one run"
    codes = torch.stack(t1[:4])
    wav = big.codec.decode_batch(codes)
    assert tuple(wav.shape) == (4, (F + 1) * 1920) and bool(torch.isfinite(wav).all())
    pre = big.codec.decode_batch(codes[:, :50])
    assert _err(wav[:, : 50 * 1920], pre) < 1e-4 * float(wav.abs().max()), "decoder is not causal in the frame index"


def test_decode_batch_in_row_chunks_equals_one_call(tts, monkeypatch):
    """Round 4: large batches are decoded in row chunks (a 64 x 400 decode in one call is slower than two 32 x 400 calls);
    every utterance's samples must not depend on the chunking (rows are independent; no few-row split-K at these sizes)."""
    rng = np.random.default_rng(91)
    codes = torch.from_numpy(rng.integers(0, 2048, size=(5, 300, 32)))
    monkeypatch.setenv("SOPRO_MIMI_CHUNK_CELLS", "100000")
    one = tts.codec.decode_batch(codes)
    monkeypatch.setenv("SOPRO_MIMI_CHUNK_CELLS", "600")  # 2 rows per chunk: 2 + 2 + 1
    chunked = tts.codec.decode_batch(codes)
    again = tts.codec.decode_batch(codes)  # (the chunk shapes replay their recorded sequences)
    assert tuple(one.shape) == (5, 300 * 1920) and torch.equal(chunked, again)
    assert float((one - chunked).abs().max()) <= 2e-6 * float(one.abs().max())  # (another tile walk for another row count: round-off class)
