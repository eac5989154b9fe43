"""Host-side audio glue of the reference-audio path (reference: src/sopro/audio.py).

File parsing, silence trimming and cropping are control logic on a few hundred kB of samples and stay on the
host (numpy); the resampler's filter bank is BUILT here and APPLIED on the GPU by ``sopro_fir1_f32``
(``MimiCodec.resample``), like every other contraction of the path.
"""
from __future__ import annotations

import math
import struct
from typing import Tuple

import numpy as np


def _load_with_optional_decoders(path: str) -> Tuple[np.ndarray, int]:
    """Anything that is not RIFF/WAVE (FLAC, OGG, MP3, ... - the demo server stores uploads under their own suffix,
    demo/server.py:93-97) goes through the decoders the reference uses when they are installed, in its order
    (src/sopro/audio.py:89-106): soundfile, then torchaudio."""
    try:
        import soundfile as sf  # noqa: PLC0415
    except Exception:  # noqa: BLE001
        sf = None
    if sf is not None:
        x, sr = sf.read(path, dtype="float32", always_2d=True)  # [N, channels]
        return np.ascontiguousarray(x.mean(axis=1, dtype=np.float32) if x.shape[1] > 1 else x[:, 0], dtype=np.float32), int(sr)
    try:
        import torchaudio  # noqa: PLC0415
    except Exception:  # noqa: BLE001
        torchaudio = None
    if torchaudio is not None:
        w, sr = torchaudio.load(path)  # [channels, N]
        w = w.float() / (2 ** 15) if str(w.dtype) == "torch.int16" else w.float()
        return np.ascontiguousarray((w.mean(dim=0) if w.shape[0] > 1 else w[0]).numpy(), dtype=np.float32), int(sr)
    raise ValueError(f"{path}: not a RIFF/WAVE file (install 'soundfile' or 'torchaudio' to read other formats)")


def load_audio_file(path: str) -> Tuple[np.ndarray, int]:
    """Audio file -> (mono float32 [N] in [-1, 1), sample rate).  reference: src/sopro/audio.py:89-106
    (soundfile ``dtype="float32"`` scaling: integer PCM / 2**(bits-1); channels averaged).  RIFF/WAVE is parsed here
    (no dependency); other containers fall back to soundfile / torchaudio when present."""
    with open(path, "rb") as f:
        data = f.read()
    if len(data) < 12 or data[:4] != b"RIFF" or data[8:12] != b"WAVE":
        return _load_with_optional_decoders(path)
    pos, fmt, pcm = 12, None, None
    while pos + 8 <= len(data):
        cid, size = data[pos:pos + 4], struct.unpack("<I", data[pos + 4:pos + 8])[0]
        body = data[pos + 8:pos + 8 + size]
        if cid == b"fmt ":
            fmt = body
        elif cid == b"data":
            pcm = body
        pos += 8 + size + (size & 1)
    if fmt is None or pcm is None or len(fmt) < 16:
        raise ValueError(f"{path}: missing fmt/data chunk")
    tag, nch, sr, _, _, bits = struct.unpack("<HHIIHH", fmt[:16])
    if tag == 0xFFFE and len(fmt) >= 26:  # WAVE_FORMAT_EXTENSIBLE: the sub-format GUID starts with the real tag
        tag = struct.unpack("<H", fmt[24:26])[0]
    if nch < 1:
        raise ValueError(f"{path}: no channels")
    if tag == 1:
        if bits == 8:
            x = (np.frombuffer(pcm, dtype=np.uint8).astype(np.float32) - 128.0) / 128.0
        elif bits == 16:
            x = np.frombuffer(pcm[: len(pcm) // 2 * 2], dtype="<i2").astype(np.float32) / 32768.0
        elif bits == 24:
            b = np.frombuffer(pcm[: len(pcm) // 3 * 3], dtype=np.uint8).reshape(-1, 3).astype(np.int32)
            v = b[:, 0] | (b[:, 1] << 8) | (b[:, 2] << 16)
            v = np.where(v >= 1 << 23, v - (1 << 24), v)
            x = v.astype(np.float32) / float(1 << 23)
        elif bits == 32:
            x = (np.frombuffer(pcm[: len(pcm) // 4 * 4], dtype="<i4").astype(np.float64) / float(1 << 31)).astype(np.float32)
        else:
            raise ValueError(f"{path}: unsupported PCM width {bits}")
    elif tag == 3:
        if bits == 32:
            x = np.frombuffer(pcm[: len(pcm) // 4 * 4], dtype="<f4").astype(np.float32)
        elif bits == 64:
            x = np.frombuffer(pcm[: len(pcm) // 8 * 8], dtype="<f8").astype(np.float32)
        else:
            raise ValueError(f"{path}: unsupported float width {bits}")
    else:
        raise ValueError(f"{path}: unsupported WAVE format tag {tag}")
    n = len(x) // nch
    x = x[: n * nch].reshape(n, nch)
    mono = x.mean(axis=1, dtype=np.float32) if nch > 1 else x[:, 0]
    return np.ascontiguousarray(mono, dtype=np.float32), int(sr)


def trim_silence_energy(wav: np.ndarray, sr: int, frame_ms: float = 25.0, hop_ms: float = 10.0, thresh_db_floor: float = -40.0,
                        prepad_ms: float = 30.0, postpad_ms: float = 30.0, min_keep_sec: float = 0.5) -> np.ndarray:
    """Energy-threshold trim of leading/trailing silence.  reference: src/sopro/audio.py:30-86 (mono [N])."""
    wav = np.asarray(wav, dtype=np.float32)
    T = int(wav.shape[-1])
    if T == 0 or T < int(sr * 0.1):
        return wav
    frame_len = max(1, int(sr * frame_ms / 1000.0))
    hop = max(1, int(sr * hop_ms / 1000.0))
    if T < frame_len:
        return wav
    n_frames = (T - frame_len) // hop + 1
    sq = wav.astype(np.float32) ** 2
    frames = np.lib.stride_tricks.as_strided(sq, shape=(n_frames, frame_len), strides=(sq.strides[0] * hop, sq.strides[0]))
    energy = frames.mean(axis=1, dtype=np.float32)
    energy_db = 10.0 * np.log10(energy + np.float32(1e-10))
    thresh_db = max(float(energy_db.max()) + thresh_db_floor, thresh_db_floor)
    voiced = np.nonzero(energy_db > thresh_db)[0]
    if voiced.size == 0:
        return wav
    start = max(0, int(voiced[0]) * hop - int(sr * prepad_ms / 1000.0))
    end = min(T, int(voiced[-1]) * hop + frame_len + int(sr * postpad_ms / 1000.0))
    if end <= start or (end - start) < int(min_keep_sec * sr):
        return wav
    return wav[start:end]


def center_crop_audio(wav: np.ndarray, win_samples: int) -> np.ndarray:
    """reference: src/sopro/audio.py:148-155"""
    T = int(wav.shape[-1])
    if win_samples <= 0 or T <= win_samples:
        return wav
    s = (T - win_samples) // 2
    return wav[..., s:s + win_samples]


def sinc_resample_bank(sr_in: int, sr_out: int, lowpass_filter_width: int = 6, rolloff: float = 0.99):
    """Polyphase Hann-windowed sinc filter bank of ``torchaudio.functional.resample`` (its defaults), the resampler
    the reference calls at src/sopro/audio.py:113-123.  Returns (bank [new, K] float32, left, orig, new): output sample
    i*new + p = sum_k bank[p, k] * x[i*orig + k - left], truncated to ceil(new * n / orig) samples."""
    g = math.gcd(int(sr_in), int(sr_out))
    orig, new = int(sr_in) // g, int(sr_out) // g
    base = min(orig, new) * rolloff
    width = math.ceil(lowpass_filter_width * orig / base)
    idx = np.arange(-width, width + orig, dtype=np.float64)[None, :] / orig
    t = np.arange(0, -new, -1, dtype=np.float64)[:, None] / new + idx
    t = np.clip(t * base, -lowpass_filter_width, lowpass_filter_width)
    window = np.cos(t * math.pi / lowpass_filter_width / 2) ** 2
    t = t * math.pi
    with np.errstate(invalid="ignore", divide="ignore"):
        k = np.where(t == 0, 1.0, np.sin(t) / t)
    k = k * window * (base / orig)
    return np.ascontiguousarray(k, dtype=np.float32), int(width), orig, new
