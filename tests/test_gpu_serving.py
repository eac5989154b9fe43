"""Batched admission (sopro_amd/serving.py): concurrent callers, mixed parameters, ragged texts / voices; every result must be
the waveform the CPU ORACLE gives for the same request (oracle/sopro_oracle.py generate_tokens + decode_full per request:
codebook 0 exact, waveform 1e-4 of peak; greedy decode: deterministic) - and the one a lone ``synthesize`` call gives."""
import threading

import numpy as np
import pytest
import torch

from conftest import assert_request_matches_oracle, oracle_request

pytestmark = pytest.mark.gpu


def test_service_batches_concurrent_requests_and_matches_single_calls(tts, cfg, mc, w, mw):
    from sopro_amd import wire
    from sopro_amd.serving import SynthesisService

    rng = np.random.default_rng(31)
    refs_tq = [torch.from_numpy(rng.integers(0, 2048, size=(n, 32))) for n in (40, 25)]
    refs = [tts.prepare_reference(ref_tokens_tq=r) for r in refs_tq]
    torch.set_num_threads(8)
    reqs = []
    for i in range(11):
        ids = torch.from_numpy(rng.integers(1, 500, size=int(rng.integers(5, 30))))
        kw = dict(max_frames=12 if i % 3 else 9, top_p=0.0, temperature=1.0, anti_loop=False)  # two parameter groups
        reqs.append((ids, refs[i % 2], kw))
    # lone calls first (the service re-partitions the chip while it is open)
    want, lone_toks, oracle = [], [], []
    for i, (ids, r, kw) in enumerate(reqs):
        toks = tts.model.generate_tokens(ids, r, **kw)
        want.append(tts.codec.decode_full(toks))
        lone_toks.append(toks)
        oracle.append(oracle_request(ids, refs_tq[i % 2], w, mw, cfg, mc, style_strength=1.0, **kw))  # generate_tokens' default strength
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
    for i, (g, wl) in enumerate(zip(got, want)):
        otoks, owav, oref = oracle[i]
        assert_request_matches_oracle(g, lone_toks[i], otoks, owav, oref, reqs[i][0], w, mw, cfg, mc, f"request {i}", style_strength=1.0, **reqs[i][2])
        assert g.shape == wl.shape
        assert float((g - wl).abs().max()) <= 1e-4 * max(1e-6, float(wl.abs().max()))
    # the byte formats a server would send
    blob = b"".join(wire.encode_stream([got[0][0]], 24000))
    sr, ch, pcm = wire.decode_stream(blob)
    assert sr == 24000 and pcm.shape[0] == got[0].shape[-1]
    with pytest.raises(RuntimeError):
        svc.submit("", refs[0], text_ids=reqs[0][0])


def test_service_continuous_mode_mixed_parameters(tts, cfg, mc, w, mw):
    """mode="continuous": every request may carry its own parameters and frame budget."""
    from sopro_amd.serving import SynthesisService

    rng = np.random.default_rng(33)
    ref_tq = torch.from_numpy(rng.integers(0, 2048, size=(30, 32)))
    ref = tts.prepare_reference(ref_tokens_tq=ref_tq)
    torch.set_num_threads(8)
    reqs = []
    for i in range(7):
        ids = torch.from_numpy(rng.integers(1, 500, size=int(rng.integers(5, 30))))
        reqs.append((ids, dict(max_frames=8 + 3 * i, top_p=0.0, temperature=0.7 + 0.1 * i, anti_loop=False)))
    lone_toks = [tts.model.generate_tokens(ids, ref, style_strength=float(tts.cfg.style_strength), **kw) for ids, kw in reqs]
    want = [tts.codec.decode_full(t) for t in lone_toks]
    oracle = [oracle_request(ids, ref_tq, w, mw, cfg, mc, **kw) for ids, kw in reqs]
    svc = SynthesisService(tts, mode="continuous", max_batch=3, ar_parts=1, ar_cus=64, max_frames=40, max_text=64, poll_every=8, bulk_batch=2)
    try:
        futs = [svc.submit("", ref, text_ids=ids, **kw) for ids, kw in reqs]
        got = [f.result(timeout=120) for f in futs]
    finally:
        svc.close()
    for i, (g, wl) in enumerate(zip(got, want)):
        otoks, owav, oref = oracle[i]
        assert_request_matches_oracle(g, lone_toks[i], otoks, owav, oref, reqs[i][0], w, mw, cfg, mc, f"request {i}", **reqs[i][1])
        assert g.shape == wl.shape and float((g - wl).abs().max()) <= 1e-4 * float(wl.abs().max())
