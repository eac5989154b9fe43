"""Byte formats either side of the path (sopro_amd/wire.py) against the reference demo server's definitions
(demo/server.py:69-143, 238-253) and a cache file written with the reference's own PreparedReference class."""
import io
import os
import pickle
import struct
import wave

import numpy as np
import pytest
import torch

from sopro_amd import wire
from sopro_amd.model import PreparedReference

GOLD = os.path.join(os.path.dirname(__file__), "golden")


def test_pcm16_conversion_truncates_like_the_reference():
    x = torch.tensor([[0.0, 0.5, -0.5, 1.0, -1.0, 1.7, -3.0, 0.99999, 3.05e-5, -3.05e-5]])
    got = np.frombuffer(wire.float_to_pcm16le(x), dtype="<i2")
    want = (x.clamp(-1.0, 1.0) * 32767.0).to(torch.int16).numpy()[0]  # demo/server.py:119-124
    assert np.array_equal(got, want)
    assert got[3] == 32767 and got[4] == -32767 and got[5] == 32767 and got[8] == 0 and got[9] == 0
    assert wire.float_to_pcm16le(torch.zeros(0)) == b""
    assert wire.float_to_pcm16le(torch.tensor([0.25, -0.25])) == struct.pack("<hh", 8191, -8191)  # 1-D input


def test_wav_bytes_are_a_mono_pcm16_riff_file():
    x = torch.sin(torch.arange(2400) * 0.05)[None] * 0.4
    data = wire.wav_bytes_from_float(x, 24000)
    with wave.open(io.BytesIO(data), "rb") as f:
        assert (f.getnchannels(), f.getsampwidth(), f.getframerate(), f.getnframes()) == (1, 2, 24000, 2400)
        assert f.readframes(2400) == wire.float_to_pcm16le(x)


def test_spro_stream_round_trip_and_errors():
    chunks = [torch.rand(1, n) * 2 - 1 for n in (11520, 0, 11520, 777)]
    blob = b"".join(wire.encode_stream(chunks, 24000))
    assert blob[:12] == b"SPRO" + struct.pack("<II", 24000, 1)  # demo/server.py:138-140
    assert struct.unpack("<I", blob[12:16])[0] == 2 * 11520     # demo/server.py:142-143; the empty chunk makes no frame
    sr, ch, pcm = wire.decode_stream(blob)
    assert (sr, ch) == (24000, 1) and pcm.shape == (11520 * 2 + 777,)
    assert pcm.tobytes() == b"".join(wire.float_to_pcm16le(c) for c in chunks)
    with pytest.raises(ValueError):
        wire.decode_stream(b"RIFF" + blob[4:])
    with pytest.raises(ValueError):
        wire.decode_stream(blob[:-3])


def test_reference_cache_written_by_the_reference_loads_here(tmp_path):
    """tests/golden/ref_cache_reference.pt was saved with sopro.model.PreparedReference (make_golden.py)."""
    ref = wire.load_reference(os.path.join(GOLD, "ref_cache_reference.pt"))
    assert isinstance(ref, PreparedReference)
    assert ref.ref_tokens_btq.shape == (1, 7, 32) and ref.ref_tokens_btq.dtype == torch.int64
    assert ref.sv_ref.shape == (1, 384) and ref.ref_seq.shape == (1, 7, 384) and len(ref.ref_kv_caches) == 3
    assert ref.ref_kv_caches[0]["k"].shape == (1, 2, 7, 192) and ref.ref_kv_caches[0]["key_padding_mask"] is None
    p = str(tmp_path / "ours.pt")
    wire.save_reference(p, ref)
    again = wire.load_reference(p)
    assert torch.equal(again.sv_ref, ref.sv_ref) and torch.equal(again.ref_kv_caches[2]["v"], ref.ref_kv_caches[2]["v"])


class _Evil:
    """A payload whose unpickling would call a builtin (what a crafted cache file does)."""

    def __reduce__(self):
        return (eval, ("__import__('os').getpid()",))


def test_reference_cache_loader_refuses_foreign_classes_and_builtins(tmp_path):
    """weights_only=True + an allow-list of exactly the PreparedReference dataclass (as demo/server.py:99,104 loads them)."""
    p = str(tmp_path / "bad.pt")
    torch.save({"x": io.BytesIO}, p)
    with pytest.raises(ValueError):
        wire.load_reference(p)
    p2 = str(tmp_path / "evil.pt")
    torch.save({"x": _Evil()}, p2)
    with pytest.raises(ValueError):
        wire.load_reference(p2)
    p3 = str(tmp_path / "notref.pt")
    torch.save({"sv_ref": torch.zeros(1)}, p3)
    with pytest.raises(ValueError):
        wire.load_reference(p3)
