"""The oracle against the REFERENCE's outputs at the shapes the benchmark quotes (tests/golden/make_golden_full.py:
S = 64 text tokens, 150-frame reference voice, 200 / 400 frames, a 160-frame stream(), the legacy cache-trim policy).
CPU only; the GPU tests compare the HIP engine with the same files."""
import numpy as np
import pytest
import torch

from conftest import golden
from oracle import sopro_oracle as O

torch.set_num_threads(8)
GREEDY = dict(top_p=0.0, temperature=1.0, anti_loop=False)


def _t(a):
    return torch.from_numpy(np.asarray(a))


@pytest.mark.parametrize("name", ["full200", "full400"])
def test_generate_and_decode_full_size(cfg, mc, w_noeos, mw, name):
    g = golden(name)
    ref = O.prepare_reference(_t(g["ref_tq"]), w_noeos, cfg)
    toks = O.generate_tokens(_t(g["ids"]), ref, w_noeos, cfg, max_frames=int(g["max_frames"]), style_strength=1.0, **GREEDY)
    want = _t(g["tokens"].astype(np.int64))
    assert tuple(toks.shape) == tuple(want.shape) == (int(g["max_frames"]) + 1, 32)
    assert torch.equal(toks[:, 0], want[:, 0])
    assert torch.equal(toks, want)
    wav = O.decode_full(toks, mw, mc).reshape(-1)
    assert float((wav - _t(g["wav"])).abs().max()) < 1e-5 * float(np.abs(g["wav"]).max())
    # the audit helpers agree with the generator on its own output (no off-argmax position)
    prep = O.prepare_conditioning(_t(g["ids"]), ref, w_noeos, cfg, max_frames=int(g["max_frames"]), style_strength=1.0)
    if name == "full200":
        assert O.ar_audit_greedy(prep, want[:, 0].tolist(), w_noeos, cfg) == (0, 0.0)
        assert O.nar_audit(prep["cond_ar"][:, : want.shape[0]], want.unsqueeze(0), w_noeos, cfg) == (0, 0.0)
        # and they do flag a wrong token
        bad = want.clone()
        bad[17, 5] = (bad[17, 5] + 1) % 2048
        n_off, gap = O.nar_audit(prep["cond_ar"][:, : want.shape[0]], bad.unsqueeze(0), w_noeos, cfg)
        assert n_off >= 1 and gap > 1e-3


def test_stream_160_frames_cache_past_the_window(cfg, mc, w_noeos, mw):
    """27 chunks of 6 frames: the codec transformer's cache passes 250 positions (DynamicSlidingWindowLayer keeps 249)."""
    g, gi = golden("stream160"), golden("full200")
    ref = O.prepare_reference(_t(gi["ref_tq"]), w_noeos, cfg)
    chunks = list(O.stream(_t(gi["ids"]), ref, w_noeos, mw, cfg, mc, max_frames=int(g["max_frames"]), style_strength=1.0, chunk_frames=6, **GREEDY))
    assert [int(c.shape[1]) for c in chunks] == g["chunk_sizes"].tolist()
    cat = torch.cat(chunks, dim=1).reshape(-1)
    assert float((cat - _t(g["stream"])).abs().max()) < 1e-5 * float(np.abs(g["stream"]).max())


def test_stream_legacy_trim_policy(cfg, mc, w_noeos, mw):
    """drop_cache_tail's legacy branch (transformers 4.57.6 cache API; src/sopro/codec/mimi.py:92-103, quirk Q6)."""
    g, gi = golden("stream_legacy"), golden("full200")
    ref = O.prepare_reference(_t(gi["ref_tq"]), w_noeos, cfg)
    kw = dict(max_frames=int(g["max_frames"]), style_strength=1.0, chunk_frames=6, **GREEDY)
    leg = torch.cat(list(O.stream(_t(gi["ids"]), ref, w_noeos, mw, cfg, mc, trim="legacy", **kw)), dim=1).reshape(-1)
    scale = float(np.abs(g["stream"]).max())
    assert float((leg - _t(g["stream"])).abs().max()) < 1e-5 * scale
    plain = torch.cat(list(O.stream(_t(gi["ids"]), ref, w_noeos, mw, cfg, mc, **kw)), dim=1).reshape(-1)
    assert float((plain - _t(g["stream"])).abs().max()) > 1e-3 * scale  # the two policies really differ


@pytest.mark.parametrize("name,cf", [("stream_c1", 1), ("stream_c16", 16)])
def test_stream_other_chunk_sizes(cfg, mc, w_noeos, mw, name, cf):
    """chunk_frames 1 and 16 against the reference's own stream() runs (tests/golden/make_golden_stream_chunks.py)."""
    g, gi = golden(name), golden("full200")
    ref = O.prepare_reference(_t(gi["ref_tq"]), w_noeos, cfg)
    chunks = list(O.stream(_t(gi["ids"]), ref, w_noeos, mw, cfg, mc, max_frames=int(g["max_frames"]), style_strength=1.0, chunk_frames=cf, **GREEDY))
    assert [int(c.shape[1]) for c in chunks] == g["chunk_sizes"].tolist()
    cat = torch.cat(chunks, dim=1).reshape(-1)
    assert float((cat - _t(g["stream"])).abs().max()) < 1e-5 * float(np.abs(g["stream"]).max())
