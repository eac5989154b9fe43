"""The oracle's Mimi decode / encode against the installed third-party ``transformers`` MimiModel run LIVE on the same
seeded synthetic checkpoint (transformers is part of the image, not of /root/reference; the stored fixtures in
tests/golden/ were produced the same way).  CPU only."""
import numpy as np
import pytest
import torch

from conftest import SEED
from oracle import sopro_oracle as O

transformers = pytest.importorskip("transformers")
torch.set_num_threads(4)


@pytest.fixture(scope="module")
def hf_and_weights(mc):
    from transformers import MimiConfig, MimiModel

    from sopro_amd.weights import synth_mimi_weights

    wnp = synth_mimi_weights(mc, SEED, with_encoder=True)
    mm = MimiModel(MimiConfig(num_quantizers=int(mc.num_quantizers))).eval()
    missing, unexpected = mm.load_state_dict({k: torch.from_numpy(v) for k, v in wnp.items()}, strict=False)
    assert not unexpected and not missing, (missing[:3], unexpected[:3])  # our layout table covers every HF tensor
    return mm, O.to_torch(wnp)


def test_decode_matches_hf_live(mc, hf_and_weights):
    mm, mw = hf_and_weights
    codes = torch.from_numpy(np.random.default_rng(8).integers(0, 2048, size=(1, 32, 5)))
    with torch.inference_mode():
        want = mm.decode(audio_codes=codes, return_dict=True).audio_values
    got = O.mimi_decode(codes, mw, mc)
    assert got.shape == want.shape
    assert float((got - want).abs().max()) < 1e-4 * float(want.abs().max())


def test_encode_matches_hf_live(mc, hf_and_weights):
    mm, mw = hf_and_weights
    wav = torch.from_numpy((0.3 * np.random.default_rng(9).standard_normal(1920 * 3 + 500)).astype(np.float32)).view(1, 1, -1)
    with torch.inference_mode():
        want = mm.encode(wav, return_dict=True).audio_codes
    got = O.mimi_encode(wav, mw, mc)
    assert torch.equal(got, want)
