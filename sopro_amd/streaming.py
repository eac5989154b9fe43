"""Chunked streaming synthesis (reference: src/sopro/streaming.py:12-152).

AR tokens are produced ``chunk_frames`` at a time by replaying the recorded per-frame hipGraph;
every chunk the NAR refiner is re-run on a left-context window of ``rf_nar`` frames and the new
frames go through the streaming Mimi decoder.  Same chunking policy, same stop rule (first EOS,
regardless of ``min_gen_frames``: streaming.py:114-115) and same yielded shapes ``[1, n*1920]``.
"""
from __future__ import annotations

from dataclasses import dataclass
from typing import Iterator, List, Optional

import torch

from .codec import MimiDecodeState, MimiStreamDecoder
from .model import PreparedReference


@dataclass
class StreamConfig:
    chunk_frames: int = 16
    nar_context_frames: Optional[int] = None
    cache_trim: str = "none"  # MimiStreamDecoder policy: "none" = transformers 5.x behaviour, "legacy" = 4.57.6 (quirk Q6)


class SoproTTSStreamer:
    def __init__(self, tts, cfg: Optional[StreamConfig] = None):
        self.tts = tts
        self.cfg = cfg or StreamConfig()
        self.mimi_stream = MimiStreamDecoder(tts.codec, trim=self.cfg.cache_trim)

    @torch.inference_mode()
    def stream(self, text: str, *, ref_audio_path: Optional[str] = None, ref_tokens_tq: Optional[torch.Tensor] = None,
               ref: Optional[PreparedReference] = None, max_frames: int = 400, top_p: float = 0.9,
               temperature: float = 1.05, anti_loop: bool = True, style_strength: Optional[float] = None,
               ref_seconds: Optional[float] = None, chunk_frames: Optional[int] = None,
               nar_context_frames: Optional[int] = None, min_gen_frames: Optional[int] = None,
               text_ids: Optional[torch.Tensor] = None, seed: Optional[int] = None) -> Iterator[torch.Tensor]:
        tts = self.tts
        model = tts.model
        ids = text_ids if text_ids is not None else tts.encode_text(text)
        if ref is None:
            ref = tts.prepare_reference(ref_audio_path=ref_audio_path, ref_tokens_tq=ref_tokens_tq, ref_seconds=ref_seconds)
        prep = model.prepare_conditioning(
            ids, ref, max_frames=max_frames,
            style_strength=float(style_strength if style_strength is not None else tts.cfg.style_strength))
        cf = int(chunk_frames if chunk_frames is not None else self.cfg.chunk_frames)
        nar_ctx = nar_context_frames if nar_context_frames is not None else self.cfg.nar_context_frames
        if nar_ctx is None:
            nar_ctx = int(model.rf_nar())
        nar_ctx = int(nar_ctx)

        hist: List[int] = []
        emitted = 0
        state = MimiDecodeState()

        def refine_and_emit(end: int) -> Optional[torch.Tensor]:
            nonlocal emitted, state
            if end <= emitted:
                return None
            ws = max(0, emitted - nar_ctx)
            cond_win = prep["cond_ar"][:, ws:end, :]
            tok_a = torch.as_tensor(hist[ws:end], dtype=torch.long).unsqueeze(0)
            toks = model.nar_refine(cond_win, tok_a).squeeze(0)
            wav, state = self.mimi_stream.decode_step(toks[emitted - ws:, :], state)
            emitted = end
            return wav if wav.numel() > 0 else None

        for _t, tok, is_eos in model.ar_stream(prep, max_frames=max_frames, top_p=top_p, temperature=temperature,
                                               anti_loop=anti_loop, min_gen_frames=min_gen_frames, lookahead=cf, seed=seed):
            if is_eos:
                break
            hist.append(int(tok))
            if len(hist) % cf == 0:
                wav = refine_and_emit(len(hist))
                if wav is not None:
                    yield wav
        if emitted < len(hist):
            wav = refine_and_emit(len(hist))
            if wav is not None:
                yield wav


@torch.inference_mode()
def stream(tts, text: str, *, ref_audio_path: Optional[str] = None, ref_tokens_tq: Optional[torch.Tensor] = None,
           ref: Optional[PreparedReference] = None, chunk_frames: int = 6, cache_trim: str = "none", **kwargs) -> Iterator[torch.Tensor]:
    """reference: src/sopro/streaming.py:133-152 (``cache_trim`` is new: see MimiStreamDecoder)"""
    streamer = SoproTTSStreamer(tts, StreamConfig(chunk_frames=chunk_frames, cache_trim=cache_trim))
    return streamer.stream(text, ref_audio_path=ref_audio_path, ref_tokens_tq=ref_tokens_tq, ref=ref,
                           chunk_frames=chunk_frames, **kwargs)
