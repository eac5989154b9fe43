"""Frame-level admission (sopro_amd/continuous.py): more requests than slots, ragged end-of-speech times, mixed frame
budgets and parameters.  Every utterance must come out as the lone ``generate_tokens`` / ``decode_full`` run gives it
(greedy decode: deterministic), whichever slot it ran in and whatever ran beside it."""
import numpy as np
import pytest
import torch

from conftest import FakeTok

pytestmark = pytest.mark.gpu


def test_continuous_admission_matches_single_runs(cfg, sopro_np, mimi_np):
    from sopro_amd import SoproTTS
    from sopro_amd.continuous import ContinuousSynthesizer

    wts = dict(sopro_np)
    hb = sopro_np["ar.head.bias"].copy()
    hb[2048] = 3.9  # end-of-speech becomes likely: utterances stop at different frames
    wts["ar.head.bias"] = hb
    tts = SoproTTS.from_weights(cfg, wts, mimi_np, FakeTok(), device="cuda:0")
    rng = np.random.default_rng(71)
    refs = [tts.prepare_reference(ref_tokens_tq=torch.from_numpy(rng.integers(0, 2048, size=(n, 32)))) for n in (22, 31, 17)]
    reqs = []
    for i in range(13):
        ids = torch.from_numpy(rng.integers(0, 512, size=int(rng.integers(5, 40))))
        reqs.append(dict(text_ids=ids, ref=refs[i % 3], max_frames=(40, 25, 33)[i % 3], top_p=0.0, temperature=(0.8, 1.0)[i % 2],
                         anti_loop=False, min_gen_frames=(6, 3)[i % 2]))
    want_tok, want_wav = [], []
    for r in reqs:
        kw = {k: v for k, v in r.items() if k not in ("text_ids", "ref")}
        toks = tts.model.generate_tokens(r["text_ids"], r["ref"], style_strength=float(cfg.style_strength), **kw)
        want_tok.append(toks)
        want_wav.append(tts.codec.decode_full(toks) if toks.shape[0] > 0 else torch.zeros(1, 1, 0, device="cuda:0"))
    lens = [int(t.shape[0]) for t in want_tok]
    assert len(set(lens)) > 3, f"fixture is not ragged: {lens}"
    eng = ContinuousSynthesizer(tts, slots=4, max_frames=40, max_text=64, poll_every=8, bulk_batch=3)
    got = eng.run(reqs)
    assert eng.stats["utterances"] == len(reqs) and eng.stats["bulk_batches"] >= 4
    for i, (g, w) in enumerate(zip(got, want_wav)):
        assert g.shape == w.shape, (i, tuple(g.shape), tuple(w.shape), lens[i])
        if w.numel():
            assert float((g - w).abs().max()) <= 1e-4 * float(w.abs().max()), i
    # a second run on the same engine (slots are reused from a dirty state)
    again = eng.run(reqs[:5])
    for g, w in zip(again, want_wav[:5]):
        assert g.shape == w.shape and (w.numel() == 0 or float((g - w).abs().max()) <= 1e-4 * float(w.abs().max()))
    assert eng.stats["slot_frames_used"] <= eng.stats["frames"] * 4
    # single requests while the engine keeps running
    f1 = eng.submit(**reqs[7])
    f2 = eng.submit(**reqs[2])
    for f, i in ((f1, 7), (f2, 2)):
        g = f.result(timeout=60)
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
        assert g.shape == w.shape and (w.numel() == 0 or float((g - w).abs().max()) <= 1e-4 * float(w.abs().max())), i
    # the engine is whole again
    t3 = tts.model.generate_tokens(reqs[0]["text_ids"], reqs[0]["ref"], style_strength=float(cfg.style_strength),
                                   **{k: v for k, v in reqs[0].items() if k not in ("text_ids", "ref")})
    assert torch.equal(t3, want_tok[0])
