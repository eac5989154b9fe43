"""Data formats either side of the synthesis path, as the reference's demo server speaks them (demo/server.py).

* PCM16 little-endian payloads and WAV byte strings (demo/server.py:119-136);
* the ``SPRO`` framed stream of ``/v1/audio/speech?stream=true`` (demo/server.py:138-143, 241-253): a 12-byte
  header ``b"SPRO" + <u32 sample rate> + <u32 channels>`` followed by frames ``<u32 length> + PCM16 bytes``;
* the cached ``PreparedReference`` files (demo/server.py:69-117): ``torch.save`` of the dataclass with CPU tensors.
  Files written by the reference pickle the class as ``sopro.model.PreparedReference``; ``load_reference`` maps that name
  onto ``sopro_amd.model.PreparedReference`` (same fields), so caches move between the two engines.

No HTTP here: transport is out of scope; these are the byte formats a server hands to its transport.
"""
from __future__ import annotations

import io
import pickle
import struct
import wave
from typing import Iterable, Iterator, Tuple

import numpy as np
import torch

from .model import PreparedReference

MAGIC = b"SPRO"


def float_to_pcm16le(wav_1xt: torch.Tensor) -> bytes:
    """demo/server.py:119-124: clamp to [-1, 1], scale by 32767, truncate toward zero, little-endian int16."""
    w = wav_1xt.detach().reshape(1, -1) if wav_1xt.ndim != 2 else wav_1xt.detach()
    w = w.to("cpu", torch.float32).clamp(-1.0, 1.0)
    return (w * 32767.0).to(torch.int16).numpy().astype("<i2", copy=False).tobytes(order="C")


def wav_bytes_from_float(wav_1xt: torch.Tensor, sr: int) -> bytes:
    """demo/server.py:126-136: a mono 16-bit RIFF/WAVE file in memory."""
    bio = io.BytesIO()
    with wave.open(bio, "wb") as wf:
        wf.setnchannels(1)
        wf.setsampwidth(2)
        wf.setframerate(int(sr))
        wf.writeframes(float_to_pcm16le(wav_1xt))
    return bio.getvalue()


def stream_header(sr: int, channels: int = 1) -> bytes:
    return MAGIC + struct.pack("<II", int(sr), int(channels))


def frame(payload: bytes) -> bytes:
    return struct.pack("<I", len(payload)) + payload


def encode_stream(chunks: Iterable[torch.Tensor], sr: int, channels: int = 1) -> Iterator[bytes]:
    """Header, then one frame per non-empty waveform chunk (what ``gen()`` of demo/server.py:238-253 yields)."""
    yield stream_header(sr, channels)
    for c in chunks:
        payload = float_to_pcm16le(c)
        if payload:
            yield frame(payload)


def decode_stream(data: bytes) -> Tuple[int, int, np.ndarray]:
    """Inverse of ``encode_stream`` over the concatenated bytes -> (sample rate, channels, int16 samples)."""
    if len(data) < 12 or data[:4] != MAGIC:
        raise ValueError("not an SPRO stream")
    sr, ch = struct.unpack("<II", data[4:12])
    pos, parts = 12, []
    while pos < len(data):
        if pos + 4 > len(data):
            raise ValueError("truncated frame header")
        (n,) = struct.unpack("<I", data[pos:pos + 4])
        pos += 4
        if pos + n > len(data) or n % 2:
            raise ValueError("truncated or odd-sized frame")
        parts.append(np.frombuffer(data[pos:pos + n], dtype="<i2"))
        pos += n
    return int(sr), int(ch), (np.concatenate(parts) if parts else np.zeros(0, dtype="<i2"))


def reference_to(ref: PreparedReference, device) -> PreparedReference:
    """demo/server.py:69-88 (_ref_to_cpu / _ref_to_device) without mutating the argument."""
    mv = lambda v: v.detach().to(device) if torch.is_tensor(v) else v  # noqa: E731
    return PreparedReference(ref_tokens_btq=mv(ref.ref_tokens_btq), sv_ref=mv(ref.sv_ref), ref_seq=mv(ref.ref_seq),
                             ref_kv_caches=[{k: mv(v) for k, v in d.items()} for d in ref.ref_kv_caches])


def save_reference(path: str, ref: PreparedReference) -> None:
    """The cache file of demo/server.py:112: ``torch.save`` of the dataclass holding CPU tensors."""
    torch.save(reference_to(ref, "cpu"), path)


def load_reference(path: str, device="cpu") -> PreparedReference:
    """A cache file written by this engine or by the reference (class ``sopro.model.PreparedReference``).

    Loaded with ``weights_only=True`` like the reference does (demo/server.py:99,104): torch's restricted unpickler plus
    an allow-list of exactly one extra global, the ``PreparedReference`` dataclass, under this package's name and under the
    name the reference pickles it with (both resolve to the class of ``sopro_amd.model``: same fields, nothing is
    imported).  Anything else in the file - ``builtins.eval``, ``torch.hub.load``, ... - is refused by torch."""
    allowed = [PreparedReference, (PreparedReference, "sopro.model.PreparedReference")]
    try:
        with torch.serialization.safe_globals(allowed):
            obj = torch.load(path, map_location="cpu", weights_only=True)
    except pickle.UnpicklingError as e:
        raise ValueError(f"{path}: refusing to load ({str(e).splitlines()[-1] if str(e) else e})") from e
    if not isinstance(obj, PreparedReference):
        raise ValueError(f"{path} does not hold a PreparedReference")
    return reference_to(obj, device)
