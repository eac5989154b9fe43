"""Shared fixtures.  ``-m "not gpu"`` runs on the CPU-only build container; ``-m gpu`` needs an MI355X."""
import os
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, "tests", "golden")

VOCAB = 512   # tests/golden/make_golden.py
SEED = 1234


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with `-m gpu` through gpurun)")


def pytest_collection_modifyitems(config, items):
    if torch.cuda.is_available():
        return
    skip = pytest.mark.skip(reason="no HIP device in this container")
    for it in items:
        if "gpu" in it.keywords:
            it.add_marker(skip)


def golden(name):
    return np.load(os.path.join(GOLDEN, name + ".npz"), allow_pickle=False)


@pytest.fixture(scope="session")
def cfg():
    from sopro_amd.config import SoproTTSConfig
    return SoproTTSConfig()


@pytest.fixture(scope="session")
def mc():
    from sopro_amd.config import MimiDecoderConfig
    return MimiDecoderConfig()


@pytest.fixture(scope="session")
def sopro_np(cfg):
    from sopro_amd.weights import synth_sopro_weights
    return synth_sopro_weights(cfg, VOCAB, SEED)


@pytest.fixture(scope="session")
def mimi_np(mc):
    from sopro_amd.weights import synth_mimi_weights
    return synth_mimi_weights(mc, SEED)


@pytest.fixture(scope="session")
def w(sopro_np):
    from oracle import sopro_oracle as O
    return O.to_torch(sopro_np)


@pytest.fixture(scope="session")
def mw(mimi_np):
    from oracle import sopro_oracle as O
    return O.to_torch(mimi_np)


class FakeTok:
    vocab_size = VOCAB

    def __init__(self):
        self.table = {}

    def encode(self, text):
        return list(self.table[text])


@pytest.fixture(scope="session")
def tts(cfg, sopro_np, mimi_np):
    """The engine under test on cuda:0 (gpu tests only)."""
    from sopro_amd import SoproTTS
    return SoproTTS.from_weights(cfg, sopro_np, mimi_np, FakeTok(), device="cuda:0")


@pytest.fixture(scope="session")
def sopro_np_noeos(cfg):
    """The checkpoint of the full-size fixtures (tests/golden/make_golden_full.py): EOS logit biased away, fixed lengths."""
    from sopro_amd.weights import synth_sopro_weights
    return synth_sopro_weights(cfg, VOCAB, SEED, suppress_eos=True)


@pytest.fixture(scope="session")
def w_noeos(sopro_np_noeos):
    from oracle import sopro_oracle as O
    return O.to_torch(sopro_np_noeos)


@pytest.fixture(scope="session")
def tts_noeos(cfg, sopro_np_noeos, mimi_np):
    """Engine with the EOS-suppressed checkpoint (gpu tests at the BASELINE shapes)."""
    from sopro_amd import SoproTTS
    return SoproTTS.from_weights(cfg, sopro_np_noeos, mimi_np, FakeTok(), device="cuda:0")


def oracle_request(ids_1d, ref_tq, w, mw, cfg, mc, **kw):
    """One request through the CPU oracle (oracle/sopro_oracle.py: generate_tokens + decode_full) -> (tokens [T, Q], wav
    [1, 1, N]).  The serving / admission tests take their expected results from here, not from the engine's own lone runs."""
    from oracle import sopro_oracle as O
    kw.setdefault("style_strength", float(cfg.style_strength))
    oref = O.prepare_reference(ref_tq, w, cfg)
    toks = O.generate_tokens(ids_1d, oref, w, cfg, **kw)
    wav = O.decode_full(toks, mw, mc) if toks.shape[0] > 0 else torch.zeros(1, 1, 0)
    return toks, wav, oref


def assert_request_matches_oracle(got_wav, got_toks, want_toks, want_wav, oref, ids_1d, w, mw, cfg, mc, what="", **kw):
    """Codebook 0 exact; refined codebooks exact or - on these random inputs - every deviating decision an audited near-tie of
    the oracle (teacher-forced logit gap < 1e-4); waveform within 1e-4 of the peak of the oracle's decode of the same tokens."""
    from oracle import sopro_oracle as O
    if got_toks is not None:
        g, wt = got_toks.cpu().long(), want_toks.cpu().long()
        assert tuple(g.shape) == tuple(wt.shape), (what, tuple(g.shape), tuple(wt.shape))
        assert torch.equal(g[:, 0], wt[:, 0]), f"{what}: codebook-0 tokens differ from the oracle's"
        if not torch.equal(g, wt):
            kw.setdefault("style_strength", float(cfg.style_strength))
            prep = O.prepare_conditioning(ids_1d, oref, w, cfg, max_frames=kw["max_frames"], style_strength=kw["style_strength"])
            n_off, gap = O.nar_audit(prep["cond_ar"][:, : g.shape[0]], g.unsqueeze(0), w, cfg)
            assert gap < 1e-4, f"{what}: {n_off} refined tokens off the oracle's arg-max, worst gap {gap:.3e}"
            want_wav = O.decode_full(g, mw, mc)
    assert tuple(got_wav.shape) == tuple(want_wav.shape), (what, tuple(got_wav.shape), tuple(want_wav.shape))
    if want_wav.numel():
        err = float((got_wav.detach().float().cpu() - want_wav).abs().max())
        assert err <= 1e-4 * float(want_wav.abs().max()), (what, err)
