"""Frame-level admission (sopro_amd/continuous.py): more requests than slots, ragged end-of-speech times, mixed frame
budgets and parameters.  Every utterance must come out as the CPU ORACLE gives it for the same request (greedy decode:
deterministic; oracle/sopro_oracle.py generate_tokens + decode_full per request: codebook 0 exact, waveform 1e-4 of peak),
whichever slot it ran in and whatever ran beside it - and, as a second check, as the engine's own lone run gives it."""
import numpy as np
import pytest
import torch

from conftest import FakeTok, assert_request_matches_oracle, oracle_request
from oracle import sopro_oracle as O

pytestmark = pytest.mark.gpu


def test_continuous_admission_matches_single_runs(cfg, mc, sopro_np, mimi_np, mw):
    from sopro_amd import SoproTTS
    from sopro_amd.continuous import ContinuousSynthesizer

    wts = dict(sopro_np)
    hb = sopro_np["ar.head.bias"].copy()
    hb[2048] = 3.9  # end-of-speech becomes likely: utterances stop at different frames
    wts["ar.head.bias"] = hb
    tts = SoproTTS.from_weights(cfg, wts, mimi_np, FakeTok(), device="cuda:0")
    rng = np.random.default_rng(71)
    refs_tq = [torch.from_numpy(rng.integers(0, 2048, size=(n, 32))) for n in (22, 31, 17)]
    refs = [tts.prepare_reference(ref_tokens_tq=r) for r in refs_tq]
    w_o = O.to_torch(wts)
    torch.set_num_threads(8)
    reqs = []
    for i in range(13):
        ids = torch.from_numpy(rng.integers(0, 512, size=int(rng.integers(5, 40))))
        reqs.append(dict(text_ids=ids, ref=refs[i % 3], max_frames=(40, 25, 33)[i % 3], top_p=0.0, temperature=(0.8, 1.0)[i % 2],
                         anti_loop=False, min_gen_frames=(6, 3)[i % 2]))
    want_tok, want_wav, oracle = [], [], []
    for i, r in enumerate(reqs):
        kw = {k: v for k, v in r.items() if k not in ("text_ids", "ref")}
        toks = tts.model.generate_tokens(r["text_ids"], r["ref"], style_strength=float(cfg.style_strength), **kw)
        want_tok.append(toks)
        want_wav.append(tts.codec.decode_full(toks) if toks.shape[0] > 0 else torch.zeros(1, 1, 0, device="cuda:0"))
        # the expected result of the request: the oracle's, not the engine's
        otoks, owav, oref = oracle_request(r["text_ids"], refs_tq[i % 3], w_o, mw, cfg, mc, **kw)
        oracle.append((otoks, owav, oref, kw))
        assert_request_matches_oracle(want_wav[-1], toks, otoks, owav, oref, r["text_ids"], w_o, mw, cfg, mc, f"lone run {i}", **kw)
    lens = [int(t.shape[0]) for t in want_tok]
    assert len(set(lens)) > 3, f"fixture is not ragged: {lens}"

    def check_oracle(i, g, what):
        otoks, owav, oref, kw = oracle[i]
        assert_request_matches_oracle(g, want_tok[i], otoks, owav, oref, reqs[i]["text_ids"], w_o, mw, cfg, mc, f"{what} {i}", **kw)
    eng = ContinuousSynthesizer(tts, slots=4, max_frames=40, max_text=64, poll_every=8, bulk_batch=3)
    got = eng.run(reqs)
    assert eng.stats["utterances"] == len(reqs) and eng.stats["bulk_batches"] >= 4
    for i, (g, w) in enumerate(zip(got, want_wav)):
        check_oracle(i, g, "slot run")
        assert g.shape == w.shape, (i, tuple(g.shape), tuple(w.shape), lens[i])
        if w.numel():
            assert float((g - w).abs().max()) <= 1e-4 * float(w.abs().max()), i
    # a second run on the same engine (slots are reused from a dirty state)
    again = eng.run(reqs[:5])
    for i, (g, w) in enumerate(zip(again, want_wav[:5])):
        check_oracle(i, g, "second run")
        assert g.shape == w.shape and (w.numel() == 0 or float((g - w).abs().max()) <= 1e-4 * float(w.abs().max()))
    assert eng.stats["slot_frames_used"] <= eng.stats["frames"] * 4
    # single requests while the engine keeps running
    f1 = eng.submit(**reqs[7])
    f2 = eng.submit(**reqs[2])
    for f, i in ((f1, 7), (f2, 2)):
        g = f.result(timeout=60)
        check_oracle(i, g, "submit")
        assert g.shape == want_wav[i].shape and (want_wav[i].numel() == 0 or float((g - want_wav[i]).abs().max()) <= 1e-4 * float(want_wav[i].abs().max()))
    eng.close()
    with pytest.raises(RuntimeError):
        eng.submit(**reqs[0])
    # the same with the chip partitioned (generation on 64 CUs, preparation / refinement / decoding on the rest)
    eng2 = ContinuousSynthesizer(tts, slots=3, max_frames=40, max_text=64, poll_every=8, bulk_batch=4, prep_batch=3, ar_cus=64,
                                 generators=2)
    try:
        got2 = eng2.run(reqs)
    finally:
        eng2.close()
    for i, (g, w) in enumerate(zip(got2, want_wav)):
        check_oracle(i, g, "partitioned run")
        assert g.shape == w.shape and (w.numel() == 0 or float((g - w).abs().max()) <= 1e-4 * float(w.abs().max())), i
    # the engine is whole again
    t3 = tts.model.generate_tokens(reqs[0]["text_ids"], reqs[0]["ref"], style_strength=float(cfg.style_strength),
                                   **{k: v for k, v in reqs[0].items() if k not in ("text_ids", "ref")})
    assert torch.equal(t3, want_tok[0])
