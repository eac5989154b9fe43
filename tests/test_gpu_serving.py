"""Batched admission (sopro_amd/serving.py): concurrent callers, mixed parameters, ragged texts / voices; every result must be
the waveform a lone ``synthesize`` call gives for the same request (greedy decode: deterministic)."""
import threading

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def test_service_batches_concurrent_requests_and_matches_single_calls(tts):
    from sopro_amd import wire
    from sopro_amd.serving import SynthesisService

    rng = np.random.default_rng(31)
    refs = [tts.prepare_reference(ref_tokens_tq=torch.from_numpy(rng.integers(0, 2048, size=(n, 32)))) for n in (40, 25)]
    reqs = []
    for i in range(11):
        ids = torch.from_numpy(rng.integers(1, 500, size=int(rng.integers(5, 30))))
        kw = dict(max_frames=12 if i % 3 else 9, top_p=0.0, temperature=1.0, anti_loop=False)  # two parameter groups
        reqs.append((ids, refs[i % 2], kw))
    # lone calls first (the service re-partitions the chip while it is open)
    want = []
    for ids, r, kw in reqs:
        toks = tts.model.generate_tokens(ids, r, **kw)
        want.append(tts.codec.decode_full(toks))
    svc = SynthesisService(tts, max_batch=4, max_wait_ms=20.0, lanes=2, ar_cus=64, ar_parts=1, ar_shared=False)
    try:
        futs = [None] * len(reqs)

        def client(i):
            ids, r, kw = reqs[i]
            futs[i] = svc.submit("", r, text_ids=ids, **kw)

        ths = [threading.Thread(target=client, args=(i,)) for i in range(len(reqs))]
        for t in ths:
            t.start()
        for t in ths:
            t.join()
        got = [f.result(timeout=120) for f in futs]
        assert svc.stats["requests"] == len(reqs) and svc.stats["batches"] < len(reqs)  # something was actually batched
    finally:
        svc.close()
    for g, w in zip(got, want):
        assert g.shape == w.shape
        assert float((g - w).abs().max()) <= 1e-4 * max(1e-6, float(w.abs().max()))
    # the byte formats a server would send
    blob = b"".join(wire.encode_stream([got[0][0]], 24000))
    sr, ch, pcm = wire.decode_stream(blob)
    assert sr == 24000 and pcm.shape[0] == got[0].shape[-1]
    with pytest.raises(RuntimeError):
        svc.submit("", refs[0], text_ids=reqs[0][0])


def test_service_continuous_mode_mixed_parameters(tts):
    """mode="continuous": every request may carry its own parameters and frame budget."""
    from sopro_amd.serving import SynthesisService

    rng = np.random.default_rng(33)
    ref = tts.prepare_reference(ref_tokens_tq=torch.from_numpy(rng.integers(0, 2048, size=(30, 32))))
    reqs = []
    for i in range(7):
        ids = torch.from_numpy(rng.integers(1, 500, size=int(rng.integers(5, 30))))
        reqs.append((ids, dict(max_frames=8 + 3 * i, top_p=0.0, temperature=0.7 + 0.1 * i, anti_loop=False)))
    want = [tts.codec.decode_full(tts.model.generate_tokens(ids, ref, style_strength=float(tts.cfg.style_strength), **kw)) for ids, kw in reqs]
    svc = SynthesisService(tts, mode="continuous", max_batch=3, ar_parts=1, ar_cus=64, max_frames=40, max_text=64, poll_every=8, bulk_batch=2)
    try:
        futs = [svc.submit("", ref, text_ids=ids, **kw) for ids, kw in reqs]
        got = [f.result(timeout=120) for f in futs]
    finally:
        svc.close()
    for g, w in zip(got, want):
        assert g.shape == w.shape and float((g - w).abs().max()) <= 1e-4 * float(w.abs().max())
