"""Mimi codec, decode side, on MI355X: the host-side mirror of the reference's ``MimiCodec`` /
``MimiStreamDecoder`` (reference: src/sopro/codec/mimi.py:18-181), whose arithmetic lives in the
third-party HuggingFace ``MimiModel`` (HF:modeling_mimi.py:1388-1406 ``_decode_frame``).

Decoding (the hot path), all channels-last fp32 on HIP kernels, is ONE call into the library per batch
(``sopro_mimi_decode`` / ``sopro_mimi_decode_stream``, csrc/stages.hip):
  tokens [B, T, 32] -> RVQ gather-sum (semantic | acoustic, HF:1128-1137) -> output projections as
  one K=512 contraction -> depthwise ConvTranspose upsample x2 (HF:1208-1216) -> 8 pre-LN transformer
  layers with RoPE and a causal sliding window of 250 (HF:729-928) -> SEANet decoder (HF:931-961, 408-447):
  overlapping-row contractions on the three-pass split-bf16 path, the 128-channel residual block, the last
  transposed convolution and the 24 kHz tail as fused weight-stationary kernels.
This module stages the codes, records / replays the call per (B, T) shape and keeps the streaming state.

Decoding is the hot path.  Encoding a reference WAV (``encode_file`` / ``encode_waveform``) runs the
Mimi encoder through the same kernels: first conv and resampler = ``sopro_fir1_f32``; residual blocks, strided
convs, transformer, downsample and the RVQ nearest-code scores = ``sopro_gemm_f32``; assignment = ``sopro_rvq_assign_f32``.
"""
from __future__ import annotations

import os
from dataclasses import dataclass
from typing import Dict, List, Optional, Sequence, Tuple

import numpy as np
import torch

from . import hip
from .config import MimiDecoderConfig
from .model import Workspace
from .pack import pack_mimi, rope_tables


@dataclass
class MimiDecodeState:
    """reference: src/sopro/codec/mimi.py:75-80 (kv = decoder-transformer cache).  The cache itself is the C side's
    ``sopro_mimi_stream_state`` (post-RoPE (k | v) rows per layer in a device buffer of ``cap_rows`` rows)."""

    cst: Optional["hip.MimiStreamState"] = None
    kv_buf: Optional[torch.Tensor] = None
    frames_seen: int = 0
    samples_emitted: int = 0
    tail_codes_tq: Optional[torch.Tensor] = None

    @property
    def kv_len(self) -> int:
        return int(self.cst.kv_len) if self.cst is not None else 0

    @property
    def pos(self) -> int:  # transformer position of the next row
        return int(self.cst.pos) if self.cst is not None else 0

    @property
    def evict(self) -> bool:  # sliding-window layers drop all but the last window-1 rows; False after a legacy-policy trim
        return bool(self.cst.evict) if self.cst is not None else True


class MimiCodec:
    def __init__(self, weights: Dict[str, "np.ndarray"], mc: Optional[MimiDecoderConfig] = None, device: str = "cuda:0",
                 precision: str = "f32"):
        if precision not in ("f32", "bf16"):
            raise ValueError("precision must be 'f32' or 'bf16'")
        self.precision = precision  # "bf16": decoder contractions with bf16 operands, one MFMA pass (fp32 accumulate / activations)
        hip.load()
        if not torch.cuda.is_available():
            raise hip.SoproHipError("no HIP device visible: the Mimi decoder has no CPU fallback")
        self.mc = mc or MimiDecoderConfig()
        self.device = torch.device(device)
        self.w = {k: v.to(self.device) for k, v in pack_mimi(weights, self.mc).items()}
        self.final_bias = float(self.w["sea.final.b"].item())
        self.ws = Workspace(self.device)
        self.stream = torch.cuda.Stream(device=self.device)
        self.num_quantizers = int(self.mc.num_quantizers)
        self.ws_budget = int(os.environ.get("SOPRO_WS_BUDGET_GB", "32")) << 30  # scratch kept per batch shape, per engine
        self.use_graph = os.environ.get("SOPRO_NO_BULK_GRAPH", "0") != "1"
        self._graphs = hip.GraphCache("mimi_graph", cap=32)  # recorded decode calls per (B, T)
        self.ws.on_clear.append(self._graphs.clear)  # (the recorded calls point into the scratch)
        self.stream_cap_rows = 1024  # cache rows of a streaming state (MimiStreamDecoder sets it per policy)
        self._rope: Optional[Tuple[torch.Tensor, torch.Tensor]] = None
        self._rope_n = 0
        self._rope_old: List[Tuple[torch.Tensor, torch.Tensor]] = []  # outgrown tables stay alive: recorded graphs point at them
        self._banks: Dict[Tuple[int, int], tuple] = {}
        # the stage engine (csrc/stages.hip): the decoder's launch sequence and its packed operands live in the library
        from .stages import codec_engine

        self.eng = codec_engine(self) if "rvq_proj.w" in self.w else None

    def on_stream(self):
        self.stream.wait_stream(torch.cuda.current_stream(self.device))
        return torch.cuda.stream(self.stream)

    def clone_lane(self) -> "MimiCodec":
        """A second decoder over the SAME device weights with its own stream and scratch buffers (pipelining)."""
        import copy

        other = copy.copy(self)
        other.ws = Workspace(self.device)
        other.stream = torch.cuda.Stream(device=self.device)
        other._graphs = hip.GraphCache("mimi_graph", cap=32)
        other.ws.on_clear.append(other._graphs.clear)
        return other

    def share_scratch(self, ws: Workspace) -> None:
        """Decode in ANOTHER decoder's scratch buffers.  For a scheduler whose decode phases never overlap (PipelinedSynthesizer with one
        throughput slot): a 64 x 200-frame call needs 12.7 GB of scratch, four lanes held it four times (71 of the 87 GiB a pipelined
        bench run peaked at, round 5).  Dropping the buffers drops every sharer's recorded calls (Workspace.on_clear)."""
        self._graphs.clear()
        self.ws = ws
        if self._graphs.clear not in ws.on_clear:
            ws.on_clear.append(self._graphs.clear)

    def _rope_tables(self, n: int) -> Tuple[torch.Tensor, torch.Tensor]:
        """cos / sin rows for positions [0, n).  Allocated once for 8192 positions (2 MB; a 400-frame stream() reaches ~1100,
        a 40 s reference ~1000); a table that is ever outgrown is kept alive next to its replacement, because the launch
        sequences recorded for synthesize() hold raw pointers into it."""
        if self._rope is None or self._rope_n < n:
            n2 = max(8192, 2 * n)
            c, s = rope_tables(n2, int(self.mc.head_dim), float(self.mc.rope_theta))
            if self._rope is not None:
                self._rope_old.append(self._rope)
            self._rope = (c.to(self.device), s.to(self.device))
            self._rope_n = n2
        return self._rope

    # ------------------------------------------------------------------ reference API
    def encode_file(self, wav_path: str, *, crop_seconds: Optional[float] = None) -> torch.Tensor:
        """Reference WAV -> Mimi codes [T, Q] (int64, on the device).  reference: src/sopro/codec/mimi.py:42-63:
        load (mono) -> energy trim -> resample to the codec rate -> optional centre crop -> MimiModel.encode."""
        from . import audio

        wav, sr = audio.load_audio_file(wav_path)
        wav = audio.trim_silence_energy(wav, sr)
        sr_t = int(self.mc.sampling_rate)
        x = self.resample(torch.from_numpy(np.ascontiguousarray(wav)).to(self.device), sr, sr_t)
        if crop_seconds is not None and crop_seconds > 0:
            fps = float(self.mc.frame_rate)
            hop = int(round(sr_t / fps))
            win_frames = max(1, int(round(crop_seconds * fps)))
            x = audio.center_crop_audio(x, win_frames * hop).contiguous()
        return self.encode_waveform(x)

    @torch.inference_mode()
    def resample(self, wav_n: torch.Tensor, sr_in: int, sr_out: int) -> torch.Tensor:
        """[N] device waveform -> [ceil(N * sr_out / sr_in)]: polyphase windowed-sinc bank on the GPU
        (reference: src/sopro/audio.py:113-123 -> torchaudio.functional.resample defaults)."""
        from . import audio

        wav_n = wav_n.to(self.device, torch.float32).contiguous()
        if int(sr_in) == int(sr_out) or wav_n.numel() == 0:
            return wav_n
        key = (int(sr_in), int(sr_out))
        if key not in self._banks:
            bank, left, orig, new = audio.sinc_resample_bank(*key)
            self._banks[key] = (torch.from_numpy(bank).to(self.device), left, orig, new)
        bank, left, orig, new = self._banks[key]
        n = int(wav_n.numel())
        n_blk = n // orig + 1  # torchaudio pads (width, width + orig): floor((n + orig - orig) / orig) + 1 blocks
        out = torch.empty(n_blk * new, device=self.device)
        with self.on_stream():
            hip.fir1(wav_n, bank, out, B=1, n_in=n, n_out=n_blk, C_=new, K=int(bank.shape[1]), stride=orig, left=left)
        self.stream.synchronize()
        return out[: -(-new * n // orig)].contiguous()

    @torch.inference_mode()
    def encode_waveform(self, wav: torch.Tensor) -> torch.Tensor:
        """[N] / [1, N] / [B, N] waveform(s) at the codec rate -> codes [T, Q] (or [B, T, Q]), T = ceil(N / 1920).
        HF:modeling_mimi.py MimiModel._encode_frame: SEANet encoder -> transformer -> stride-2 downsample -> split residual VQ;
        the launch sequence is ``sopro_mimi_encode`` (csrc/stages.hip), every contraction in exact fp32."""
        if "enc.conv0.w" not in self.w:
            raise hip.SoproHipError("this Mimi checkpoint was loaded without its encoder-side tensors")
        squeeze = wav.dim() == 1
        x = (wav.unsqueeze(0) if squeeze else wav.reshape(-1, wav.shape[-1])).to(self.device, torch.float32).contiguous()
        B, N = int(x.shape[0]), int(x.shape[1])
        if N == 0:
            raise ValueError("empty waveform")
        lib, Q = hip.load(), self.num_quantizers
        T = -(-N // int(self.mc.frame_samples))
        with self.on_stream():
            codes = torch.empty(B * T, Q, dtype=torch.int32, device=self.device)
            wsb = torch.empty(int(lib.sopro_mimi_encode_workspace_bytes(self.eng.h, B, N)) // 4 + 64, device=self.device)  # per call: lengths vary freely
            hip._check(lib.sopro_mimi_encode(self.eng.h, hip.ptr(wsb), hip.ptr(x), B, N, hip.ptr(codes, torch.int32), hip._stream()), "sopro_mimi_encode")
        self.stream.synchronize()
        out = codes.view(B, T, Q).long()
        return out[0] if squeeze else out

    @torch.inference_mode()
    def decode_full(self, codes_tq: torch.Tensor) -> torch.Tensor:
        """[T, Q] -> [1, 1, T*1920]  (reference: src/sopro/codec/mimi.py:65-72)."""
        wav = self.decode_batch(codes_tq.unsqueeze(0))
        return wav.unsqueeze(1)

    @torch.inference_mode()
    def decode_batch(self, codes_btq: torch.Tensor, state: Optional[MimiDecodeState] = None,
                     timings: Optional[Dict[str, float]] = None) -> torch.Tensor:
        """[B, T, Q] integer codes -> [B, T*1920] fp32 waveform: ``sopro_mimi_decode`` (csrc/stages.hip), recorded per (B, T)
        shape and replayed.  With ``state`` (B == 1) the transformer attends over the cached keys / values of earlier calls, as
        MimiModel.decode(decoder_past_key_values=...): ``sopro_mimi_decode_stream``."""
        mc, dev, ws = self.mc, self.device, self.ws
        B, T, Q = codes_btq.shape
        if Q != self.num_quantizers:
            raise ValueError(f"expected {self.num_quantizers} codebooks, got {Q}")
        if T == 0:
            return torch.zeros(B, 0, device=dev)
        if state is not None and B != 1:
            raise ValueError("streaming decode state is single-utterance")
        if self.eng is None:
            raise hip.SoproHipError("this Mimi checkpoint was loaded without its decoder-side tensors")
        lib, eng = hip.load(), self.eng
        if ws.over(self.ws_budget):  # many batch shapes seen: start over, unless THIS call's chunk shapes all have their recorded sequences
            r0 = int(lib.sopro_mimi_chunk_rows(B, T)) if state is None else B
            need = {(min(B, b0 + r0) - b0, T) for b0 in range(0, B, r0)}
            if not need <= {(k[0], k[1]) for k in self._graphs.graphs}:
                torch.cuda.synchronize(self.device)
                self._graphs.clear()
                ws.clear()
        # Large batches are decoded in balanced row chunks of ~12800 frames (32 x 400, 64 x 200): a 64 x 400 decode in one call
        # measured slower per utterance than two 32 x 400 calls (39.5 vs 33.9 ms per 32; its scratch is 148 GB) - with chunks a
        # scheduler can still coalesce two long-form jobs into one 64-row generation / refinement pass (profiles/r04_experiments.md).
        # The rule lives in the library (sopro_mimi_chunk_rows; sopro_mimi_decode applies it for hosts without this class).
        rows = int(lib.sopro_mimi_chunk_rows(B, T)) if state is None else B
        chunks = [(b0, min(B, b0 + rows)) for b0 in range(0, B, rows)]
        hop = int(mc.frame_samples)
        with self.on_stream():
            # The codes are read WHERE THEY ARE when they are the refinement's own int32 matrix on this device (a scheduler's pass:
            # SoproTTSModel.phase_nar(raw=True)); anything else is staged once into a persistent buffer.
            src = codes_btq
            if not (src.is_cuda and src.device == dev and src.dtype == torch.int32 and src.is_contiguous()):
                src = ws.get(f"rvq.tok.{B}x{T}", (B, T, Q), dtype=torch.int32)
                src.copy_(codes_btq.to(dev))
            # The result is a fresh tensor the caller owns, and the decoder's LAST launch (the only one that touches it) writes
            # straight into it: that launch is issued here with this call's destination, everything in front of it is replayed
            # from the sequence recorded for (rows, T, source) - no `clone`, no `copy_` of a 98 MB waveform per pass (round 5).
            wav = torch.empty(B, T * hop, device=dev)
            for b0, b1 in chunks:
                Bc = b1 - b0
                tok_ptr = src.data_ptr() + b0 * T * Q * 4
                out_ptr = wav.data_ptr() + b0 * T * hop * 4
                scratch = ws.get(f"mimi.stage_ws.{Bc}x{T}", (int(lib.sopro_mimi_workspace_bytes(eng.h, Bc, T)),), dtype=torch.uint8)
                if state is None:
                    def body(tok_ptr=tok_ptr, out_ptr=out_ptr, scratch=scratch, Bc=Bc):
                        hip._check(lib.sopro_mimi_decode_parts(eng.h, scratch.data_ptr(), tok_ptr, Bc, T, out_ptr, 1, hip._stream()), "sopro_mimi_decode_parts")

                    if self.use_graph:
                        self._graphs.run((Bc, T, tok_ptr), body)
                    else:
                        body()
                    hip._check(lib.sopro_mimi_decode_parts(eng.h, scratch.data_ptr(), tok_ptr, Bc, T, out_ptr, 2, hip._stream()), "sopro_mimi_decode_parts")
                else:
                    import ctypes as C

                    if state.cst is None:  # first call of a stream: the cache buffer (window + chunk rows under the evicting policy)
                        cap = self.stream_cap_rows
                        state.kv_buf = torch.empty(int(lib.sopro_mimi_stream_kv_bytes(eng.h, cap)), dtype=torch.uint8, device=dev)
                        state.cst = hip.MimiStreamState()
                        hip._check(lib.sopro_mimi_stream_init(eng.h, C.byref(state.cst), state.kv_buf.data_ptr(), cap), "sopro_mimi_stream_init")
                    hip._check(lib.sopro_mimi_decode_stream(eng.h, scratch.data_ptr(), C.byref(state.cst), tok_ptr, T, out_ptr, hip._stream()),
                               "sopro_mimi_decode_stream")
        self.stream.synchronize()
        return wav


class MimiStreamDecoder:
    """Chunked decode with a 2-frame token overlap and a growing transformer cache
    (reference: src/sopro/codec/mimi.py:83-181).  The reference's output depends on the installed transformers (quirk Q6):
      * ``trim="none"`` (default) - transformers 5.x: ``drop_cache_tail`` finds no legacy-cache API and trims nothing
        (SURVEY.md Appendix C); the sliding-window cache layers keep the last 249 positions;
      * ``trim="legacy"`` - the lock-pinned 4.57.6: ``drop_cache_tail``'s legacy branch (:92-103) drops the last ``ov``
        cached positions of every layer (``ov`` frames are 2*ov positions: only half of the overlap goes), rebuilds the
        cache as plain layers that never evict, and the next call's positions continue from the trimmed length.
    Both are pinned by reference-generated fixtures (tests/golden/stream160.npz, stream_legacy.npz)."""

    def __init__(self, codec: MimiCodec, overlap_frames: int = 2, trim: str = "none"):
        if trim not in ("none", "legacy"):
            raise ValueError("trim must be 'none' or 'legacy'")
        self.codec = codec
        self.overlap_frames = int(overlap_frames)
        self.trim = trim

    @torch.inference_mode()
    def decode_step(self, codes_chunk_tq: torch.Tensor, state: Optional[MimiDecodeState] = None
                    ) -> Tuple[torch.Tensor, MimiDecodeState]:
        st = state or MimiDecodeState()
        hop = int(self.codec.mc.frame_samples)
        n_new = int(codes_chunk_tq.shape[0])
        if n_new == 0:
            return torch.zeros(1, 0, device=self.codec.device), st
        dev = self.codec.device
        chunk = codes_chunk_tq.to(dev).long()
        ov = 0
        codes_in = chunk
        if self.overlap_frames > 0 and st.tail_codes_tq is not None and st.tail_codes_tq.numel() > 0:
            ov = min(self.overlap_frames, int(st.tail_codes_tq.shape[0]))
            codes_in = torch.cat([st.tail_codes_tq[-ov:], chunk], dim=0)
        if self.trim == "legacy" and ov > 0 and st.cst is not None and st.kv_len > 0:
            import ctypes as C

            hip._check(hip.load().sopro_mimi_stream_trim(C.byref(st.cst), ov), "sopro_mimi_stream_trim")
        # the evicting policy keeps window - 1 rows + the call's own; the legacy policy's plain layers keep everything
        self.codec.stream_cap_rows = 4096 if self.trim == "legacy" else 1024
        wav = self.codec.decode_batch(codes_in.unsqueeze(0), state=st)
        wav = wav[:, : (ov + n_new) * hop][:, ov * hop:]
        st.frames_seen += n_new
        st.samples_emitted += int(wav.shape[1])
        keep = min(self.overlap_frames, int(codes_in.shape[0]))
        st.tail_codes_tq = codes_in[-keep:].clone() if self.overlap_frames > 0 else None
        return wav, st
