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
