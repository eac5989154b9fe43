"""``SoproTTS``: the drop-in public facade (reference: src/sopro/model.py:404-583).

Same constructor / method signatures and error behaviour as the reference class, so existing call
sites (``README.md:69-120``, ``src/sopro/cli.py``, ``demo/server.py:224,241``) keep working; the body
of every method runs on the MI355X engine (``sopro_amd.model`` / ``sopro_amd.codec``).  Additions
that the reference does not have: ``synthesize_batch`` and ``from_weights``.
"""
from __future__ import annotations

import os
from typing import Any, Dict, Iterator, List, Optional, Sequence

import numpy as np
import torch

from .codec import MimiCodec
from .config import DEFAULT_MIMI_ID, TARGET_SR, MimiDecoderConfig, SoproTTSConfig
from .model import PreparedReference, SoproTTSModel
from .weights import load_cfg_from_safetensors, load_safetensors


class SoproTTS:
    def __init__(self, model: SoproTTSModel, cfg: SoproTTSConfig, tokenizer: Any, codec: MimiCodec, device: str):
        self.model = model
        self.cfg = cfg
        self.tokenizer = tokenizer
        self.codec = codec
        self.device = torch.device(device)

    # ------------------------------------------------------------------ construction
    @classmethod
    def from_pretrained(cls, repo_id: str, *, revision: Optional[str] = None, cache_dir: Optional[str] = None,
                        token: Optional[str] = None, device: Optional[str] = None, precision: str = "f32") -> "SoproTTS":
        """reference: src/sopro/model.py:419-451.  ``repo_id`` may also be a local directory holding
        ``model.safetensors`` (+ tokenizer files) and, for the codec, ``mimi/model.safetensors``.  ``precision`` (new):
        "f32" reproduces the fp32 reference; "bf16" runs the NAR / Mimi contractions with bf16 operands."""
        device = device or "cuda"
        if os.path.isdir(repo_id):
            local_dir = repo_id
        else:
            from huggingface_hub import snapshot_download  # reference: src/sopro/hub.py:15-27

            local_dir = snapshot_download(repo_id=repo_id, revision=revision, cache_dir=cache_dir, token=token)
        model_path = os.path.join(local_dir, "model.safetensors")
        if not os.path.exists(model_path):
            raise FileNotFoundError(f"Expected {model_path} in repo snapshot.")
        cfg = load_cfg_from_safetensors(model_path)
        weights = load_safetensors(model_path)
        tokenizer = _load_tokenizer(local_dir)
        mimi_dir = os.path.join(local_dir, "mimi")
        if os.path.exists(os.path.join(mimi_dir, "model.safetensors")):
            mimi_weights = load_safetensors(os.path.join(mimi_dir, "model.safetensors"))
        else:
            from huggingface_hub import snapshot_download

            mdir = snapshot_download(repo_id=DEFAULT_MIMI_ID, cache_dir=cache_dir, token=token)
            mimi_weights = load_safetensors(os.path.join(mdir, "model.safetensors"))
        return cls.from_weights(cfg, weights, mimi_weights, tokenizer, device=device, precision=precision)

    @classmethod
    def from_weights(cls, cfg: SoproTTSConfig, weights: Dict[str, np.ndarray], mimi_weights: Dict[str, np.ndarray],
                     tokenizer: Any, *, device: str = "cuda", seed: int = 0, use_graph: bool = True,
                     precision: str = "f32") -> "SoproTTS":
        """Build from in-memory checkpoints (reference ``state_dict`` names; HF Mimi names)."""
        dev = torch.device(device)
        if dev.type == "cuda" and dev.index is None:
            dev = torch.device("cuda", torch.cuda.current_device() if torch.cuda.is_available() else 0)
        model = SoproTTSModel(cfg, weights, str(dev), seed=seed, use_graph=use_graph, precision=precision)
        codec = MimiCodec(mimi_weights, MimiDecoderConfig(num_quantizers=int(cfg.num_codebooks)), str(dev), precision=precision)
        return cls(model, cfg, tokenizer, codec, str(dev))

    # ------------------------------------------------------------------ reference API
    def encode_text(self, text: str) -> torch.Tensor:
        ids = self.tokenizer.encode(text)  # reference: src/sopro/model.py:453-455
        return torch.tensor(ids, dtype=torch.long)

    @torch.inference_mode()
    def encode_speaker(self, *, ref_audio_path: Optional[str] = None, ref_tokens_tq: Optional[torch.Tensor] = None,
                       ref_seconds: Optional[float] = None) -> torch.Tensor:
        """reference: src/sopro/model.py:457-475 -> the voice's speaker vector [sv_student_dim] (Token2SV over the cropped
        reference tokens; the same vector ``prepare_reference`` stores as ``sv_ref``)."""
        ref = self.encode_reference(ref_audio_path=ref_audio_path, ref_tokens_tq=ref_tokens_tq, ref_seconds=ref_seconds)
        ref_btq = ref.unsqueeze(0)
        lengths = torch.tensor([int(ref_btq.size(1))], dtype=torch.long)
        return self.model.token2sv(ref_btq, lengths=lengths).squeeze(0).detach()

    @torch.inference_mode()
    def encode_reference(self, *, ref_audio_path: Optional[str] = None, ref_tokens_tq: Optional[torch.Tensor] = None,
                         ref_seconds: Optional[float] = None) -> torch.Tensor:
        """reference: src/sopro/model.py:477-514 (token path only; audio -> tokens needs the Mimi encoder)."""
        if ref_tokens_tq is None and ref_audio_path is None:  # model.py:486-493
            raise RuntimeError("SoproTTS requires a reference. Provide ref_audio_path=... or ref_tokens_tq=...")
        if ref_tokens_tq is not None and ref_audio_path is not None:
            raise RuntimeError("Provide only one of ref_audio_path or ref_tokens_tq (not both).")
        if ref_seconds is None:
            ref_seconds = 12.0  # model.py:495-496
        if ref_tokens_tq is None:
            return self.codec.encode_file(ref_audio_path, crop_seconds=ref_seconds if ref_seconds > 0 else None)
        ref = ref_tokens_tq.long()
        if ref_seconds and ref_seconds > 0:
            fps = float(self.cfg.mimi_fps)
            win = max(1, int(round(ref_seconds * fps)))
            T = int(ref.shape[0])
            if T > win:  # center crop, reference: src/sopro/sampling.py:8-13
                s = (T - win) // 2
                ref = ref[s: s + win]
        return ref

    @torch.inference_mode()
    def prepare_reference(self, *, ref_audio_path: Optional[str] = None, ref_tokens_tq: Optional[torch.Tensor] = None,
                          ref_seconds: Optional[float] = None) -> PreparedReference:
        """reference: src/sopro/model.py:516-529"""
        ref = self.encode_reference(ref_audio_path=ref_audio_path, ref_tokens_tq=ref_tokens_tq, ref_seconds=ref_seconds)
        return self.model.prepare_reference(ref)

    @torch.inference_mode()
    def synthesize(self, text: str, *, ref: Optional[PreparedReference] = None, ref_audio_path: Optional[str] = None,
                   ref_tokens_tq: Optional[torch.Tensor] = None, max_frames: int = 400, top_p: float = 0.9,
                   temperature: float = 1.05, anti_loop: bool = True, style_strength: Optional[float] = None,
                   ref_seconds: Optional[float] = None, min_gen_frames: Optional[int] = None,
                   seed: Optional[int] = None) -> torch.Tensor:
        """reference: src/sopro/model.py:531-575 -> waveform [1, 1, N] on ``self.device``.  ``seed`` (new) pins the sampler's
        draws: the same seed, text and voice give the same audio; without it every call is a new take (the reference
        draws from torch's global generator; its CLI seeds that once, src/sopro/cli.py:72-75)."""
        text_ids = self.encode_text(text)
        if ref is None:
            ref = self.prepare_reference(ref_audio_path=ref_audio_path, ref_tokens_tq=ref_tokens_tq, ref_seconds=ref_seconds)
        tokens = self.model.generate_tokens(
            text_ids, ref, max_frames=max_frames, top_p=top_p, temperature=temperature, anti_loop=anti_loop,
            style_strength=float(style_strength if style_strength is not None else self.cfg.style_strength),
            min_gen_frames=min_gen_frames, seed=seed)
        return self.codec.decode_full(tokens)

    @torch.inference_mode()
    def synthesize_batch(self, texts: Sequence[str], refs: Sequence[PreparedReference], *, max_frames: int = 400,
                         top_p: float = 0.9, temperature: float = 1.05, anti_loop: bool = True,
                         style_strength: Optional[float] = None, min_gen_frames: Optional[int] = None,
                         timings: Optional[Dict[str, float]] = None, text_ids: Optional[Sequence[torch.Tensor]] = None,
                         phase_locks: Optional[tuple] = None, seed: Optional[int] = None, nonces: Optional[Sequence[int]] = None,
                         row_ids: Optional[Sequence[int]] = None) -> List[torch.Tensor]:
        """New: B utterances in one pass (batched AR graph, NAR and Mimi decode) -> list of [1, 1, N_b].
        ``nonces`` / ``row_ids``: per-utterance sampler stream of a scheduler that coalesces requests (see model._ARRun)."""
        import contextlib
        import time

        ids = list(text_ids) if text_ids is not None else [self.encode_text(t) for t in texts]
        locks = tuple(phase_locks) if phase_locks is not None else ()
        ar_lock = locks[0] if len(locks) > 0 else contextlib.nullcontext()
        bulk_lock = locks[1] if len(locks) > 1 else contextlib.nullcontext()
        cond_gate = locks[2] if len(locks) > 2 else contextlib.nullcontext()  # (a pipeline: conditioning phases take turns in job order)
        ss = float(style_strength if style_strength is not None else self.cfg.style_strength)
        from .model import _PhaseTimer

        # Every phase runs with ITS stream as the current one: stray torch ops (slices, clamps, allocations) are then queued where
        # the phase's kernels are, and no phase ever waits on a stream another engine of a pipeline may be generating on.
        self.model.prep_stream.wait_stream(torch.cuda.current_stream(self.device))  # the caller's inputs
        # conditioning needs no generation slot: it overlaps with whatever the other engines are doing
        with cond_gate, torch.cuda.stream(self.model.prep_stream):
            ev = _PhaseTimer(self.model.prep_stream, timings)
            # the stages of a pass hand their results over IN PLACE (round 5): conditioning writes into the AR plan's buffer, the
            # refinement reads that buffer and the plan's token history, the decoder reads the refinement's token matrix and writes
            # into the tensor this call returns - no copy, fill or cast of the runtime's runs between the library's launch sequences
            plan = self.model.plan_for(ids, max_frames)
            prep = self.model.phase_cond(ids, refs, max_frames=max_frames, style_strength=ss, plan=plan)
            # the AR phase's own preparation (plan buffers, folded text operands) belongs here too: the generation slot then
            # only replays frames (it sat idle for 2-3.5 ms per phase while this ran inside it)
            run = self.model.ar_prepare(prep, top_p=top_p, temperature=temperature, anti_loop=anti_loop, min_gen_frames=min_gen_frames,
                                        seed=seed, nonces=nonces, row_ids=row_ids)
            ev.mark("cond")
        with ar_lock, torch.cuda.stream(self.model.stream):  # latency-bound phase: AR graph replay (a pipeline picks the stream with the lock)
            ev = _PhaseTimer(self.model.stream, timings)
            if timings is not None:
                timings["_ar_t0"] = time.perf_counter()
            state = self.model.phase_ar(ids, refs, max_frames=max_frames, top_p=top_p, temperature=temperature, anti_loop=anti_loop,
                                        style_strength=ss, min_gen_frames=min_gen_frames, ev=ev, prep=prep, seed=seed, run=run)
            if timings is not None:
                timings["_ar_t1"] = time.perf_counter()
        # (the AR phase ends with its token history on the host side of a stream sync: the next phase needs no stream wait)
        with bulk_lock, torch.cuda.stream(self.model.bulk_stream):  # throughput-bound phase: NAR refinement + Mimi decode
            t0 = time.perf_counter()
            if timings is not None:
                timings["_bulk_t0"] = t0
            # refinement and decoding are queued back to back on the one stream (no host round trip between them: the partition
            # does not idle while the host wakes up and issues the decoder); with phase timings on, the host syncs in between
            same = self.codec.stream is self.model.bulk_stream or self.codec.stream.cuda_stream == self.model.bulk_stream.cuda_stream
            fused = same
            evs = None
            if fused and timings is not None:  # phase times from device events (a host sync between the phases is what fusing removes)
                evs = [torch.cuda.Event(enable_timing=True) for _ in range(3)]
                evs[0].record(self.model.bulk_stream)
            full = self.model.phase_nar(state, full=True, sync=not fused, raw=fused)  # [B, Tn, Q]
            if evs is not None:
                evs[1].record(self.model.bulk_stream)
            t1 = time.perf_counter()
            lens = [int(n) for n in state["lens"]]
            B, Tn = int(full.shape[0]), int(full.shape[1])
            if max(lens) == 0:
                return [torch.zeros(1, 1, 0, device=self.device) for _ in range(B)]
            # The padded batch goes to the decoder as it is: the decoder is causal, so the (valid, meaningless) codes a row holds
            # past its own length never reach the samples that are returned.
            codes = full
            wav = self.codec.decode_batch(codes)  # causal decoder: padding frames never reach earlier samples
            if fused and self.model.nar_guard(redo=True):  # (decode_batch has synchronised the stream: the pass's range word is in)
                wav = self.codec.decode_batch(codes)  # the refinement was repeated on the six-pass operands: decode its tokens
            if timings is not None:
                if evs is not None:
                    evs[2].record(self.model.bulk_stream)
                    evs[2].synchronize()
                    timings["nar"] = timings.get("nar", 0.0) + evs[0].elapsed_time(evs[1]) * 1e-3
                    timings["mimi"] = timings.get("mimi", 0.0) + evs[1].elapsed_time(evs[2]) * 1e-3
                else:
                    timings["nar"] = timings.get("nar", 0.0) + (t1 - t0)
                    timings["mimi"] = timings.get("mimi", 0.0) + (time.perf_counter() - t1)
                timings["_bulk_t1"] = time.perf_counter()
        hop = int(self.codec.mc.frame_samples)
        return [wav[b, : lens[b] * hop].reshape(1, 1, -1) for b in range(B)]

    def clone_lane(self) -> "SoproTTS":
        """Another engine over the same device weights (own streams / scratch), for pipelining batches."""
        return SoproTTS(self.model.clone_lane(), self.cfg, self.tokenizer, self.codec.clone_lane(), str(self.device))

    def stream(self, text: str, **kwargs) -> Iterator[torch.Tensor]:
        """reference: src/sopro/model.py:577-580"""
        from .streaming import stream

        return stream(self, text, **kwargs)

    def save_wav(self, path: str, wav: torch.Tensor) -> None:
        """reference: src/sopro/model.py:582-583 (16-bit PCM via the stdlib; soundfile is not required)."""
        import wave

        x = wav.detach().reshape(-1).float().clamp(-1.0, 1.0).cpu().numpy()
        pcm = (x * 32767.0).astype("<i2")
        with wave.open(path, "wb") as f:
            f.setnchannels(1)
            f.setsampwidth(2)
            f.setframerate(TARGET_SR)
            f.writeframes(pcm.tobytes())


def _load_tokenizer(local_dir: str):
    """reference: src/sopro/tokenizer.py:15-38 (HF AutoTokenizer + BOS/EOS)."""
    from transformers import AutoTokenizer

    class TextTokenizer:
        def __init__(self, d: str):
            self.tok = AutoTokenizer.from_pretrained(d, use_fast=True)
            if self.tok.pad_token_id is None:
                self.tok.add_special_tokens({"pad_token": "<|pad|>"})
            self.bos_id = int(self.tok.bos_token_id) if self.tok.bos_token_id is not None else None
            self.eos_id = int(self.tok.eos_token_id) if self.tok.eos_token_id is not None else None
            self.vocab_size = int(self.tok.vocab_size + len(self.tok.get_added_vocab()))

        def encode(self, text: str) -> List[int]:
            ids = self.tok.encode(text, add_special_tokens=False)
            if self.bos_id is not None and self.eos_id is not None:
                ids = [self.bos_id] + ids + [self.eos_id]
            return ids

    return TextTokenizer(local_dir)
