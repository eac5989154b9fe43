"""Mimi codec, decode side, on MI355X: the host-side mirror of the reference's ``MimiCodec`` /
``MimiStreamDecoder`` (reference: src/sopro/codec/mimi.py:18-181), whose arithmetic lives in the
third-party HuggingFace ``MimiModel`` (HF:modeling_mimi.py:1388-1406 ``_decode_frame``).

Pipeline, all channels-last fp32 on HIP kernels:
  tokens [B, T, 32] -> RVQ gather-sum (semantic | acoustic, HF:1128-1137) -> output projections as
  one K=512 contraction -> depthwise ConvTranspose upsample x2 (HF:1208-1216) -> 8 pre-LN transformer
  layers with RoPE and a causal sliding window of 250 (HF:729-928) -> SEANet decoder: every Conv1d /
  ConvTranspose1d is the overlapping-row contraction of ``sopro_gemm_f32`` with ELU fused on the
  operand load and the residual add fused on the store (HF:931-961, 408-447) -> last 64->1 conv.

Only decoding is on the hot path.  Encoding a reference WAV (``encode_file``) needs the Mimi encoder,
which is SURVEY.md 8(f) rank 1 ("next") and raises here.
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
    """reference: src/sopro/codec/mimi.py:75-80 (kv = decoder-transformer cache)."""

    kv: Optional[List[torch.Tensor]] = None  # per layer [len, 1024] rows of (k | v), post-RoPE
    kv_len: int = 0
    pos: int = 0  # transformer positions consumed so far
    frames_seen: int = 0
    samples_emitted: int = 0
    tail_codes_tq: Optional[torch.Tensor] = None


class MimiCodec:
    def __init__(self, weights: Dict[str, "np.ndarray"], mc: Optional[MimiDecoderConfig] = None, device: str = "cuda:0"):
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
        self.fuse_tail = os.environ.get("SOPRO_UNFUSED_TAIL", "0") != "1"
        self._rope: Optional[Tuple[torch.Tensor, torch.Tensor]] = None
        self._rope_n = 0
        Q, V = self.num_quantizers, int(self.mc.codebook_size)
        ns = int(self.mc.num_semantic_quantizers)
        i32 = lambda v: torch.tensor(list(v), dtype=torch.int32, device=self.device)  # noqa: E731
        self._sem = (i32(range(ns)), i32([q * V for q in range(ns)]), torch.ones(ns, device=self.device))
        self._ac = (i32(range(ns, Q)), i32([q * V for q in range(ns, Q)]), torch.ones(Q - ns, device=self.device))

    def on_stream(self):
        self.stream.wait_stream(torch.cuda.current_stream(self.device))
        return torch.cuda.stream(self.stream)

    def clone_lane(self) -> "MimiCodec":
        """A second decoder over the SAME device weights with its own stream and scratch buffers (pipelining)."""
        import copy

        other = copy.copy(self)
        other.ws = Workspace(self.device)
        other.stream = torch.cuda.Stream(device=self.device)
        return other

    def _rope_tables(self, n: int) -> Tuple[torch.Tensor, torch.Tensor]:
        if self._rope is None or self._rope_n < n:
            n2 = max(1024, 2 * n)
            c, s = rope_tables(n2, int(self.mc.head_dim), float(self.mc.rope_theta))
            self._rope = (c.to(self.device), s.to(self.device))
            self._rope_n = n2
        return self._rope

    # ------------------------------------------------------------------ reference API
    def encode_file(self, *a, **k):
        raise NotImplementedError("Mimi encoding of reference audio is not on the MI355X hot path yet "
                                  "(SURVEY.md 8f rank 1): pass ref_tokens_tq= or a PreparedReference")

    @torch.inference_mode()
    def decode_full(self, codes_tq: torch.Tensor) -> torch.Tensor:
        """[T, Q] -> [1, 1, T*1920]  (reference: src/sopro/codec/mimi.py:65-72)."""
        wav = self.decode_batch(codes_tq.unsqueeze(0))
        return wav.unsqueeze(1)

    @torch.inference_mode()
    def decode_batch(self, codes_btq: torch.Tensor, state: Optional[MimiDecodeState] = None,
                     timings: Optional[Dict[str, float]] = None) -> torch.Tensor:
        """[B, T, Q] integer codes -> [B, T*1920] fp32 waveform.  With ``state`` (B == 1) the transformer
        attends over the cached keys/values of earlier calls, as MimiModel.decode(decoder_past_key_values=...)."""
        mc, w, dev, ws = self.mc, self.w, self.device, self.ws
        B, T, Q = codes_btq.shape
        if Q != self.num_quantizers:
            raise ValueError(f"expected {self.num_quantizers} codebooks, got {Q}")
        if T == 0:
            return torch.zeros(B, 0, device=dev)
        HS, CD = int(mc.hidden_size), int(mc.codebook_dim)
        N2 = 2 * T
        H, dh = int(mc.num_attention_heads), int(mc.head_dim)
        win = int(mc.sliding_window)
        PADX = int(mc.kernel_size) - 1  # 6 zero rows in front of the first SEANet conv's input
        with self.on_stream():
            tok = codes_btq.to(dev).to(torch.int32).contiguous().view(B * T, Q)
            # ---- RVQ decode + output projections (HF:modeling_mimi.py:1128-1137)
            emb = ws.get("rvq.emb", (B * T, 2 * CD))
            hip.codebook_sum(tok, Q, *self._sem, w["codebooks"], emb, rows=B * T, D=CD, ldo=2 * CD)
            hip.codebook_sum(tok, Q, *self._ac, w["codebooks"], emb, rows=B * T, D=CD, ldo=2 * CD, o_off=CD)
            q = ws.get("rvq.q", (B * T, HS))
            hip.gemm(emb, w["rvq_proj.w"], q, M=B * T, N=HS, K=2 * CD)
            # ---- upsample into the (zero-padded) transformer stream
            xs_stride = (PADX + N2) * HS
            X = ws.get("tr.x", (B, PADX + N2, HS), zero=True)
            hip.upsample2(q, w["upsample.w"], X, B=B, T=T, C_=HS, y_seg_stride=xs_stride, y_off=PADX * HS)
            # ---- transformer
            past = 0
            if state is not None:
                if B != 1:
                    raise ValueError("streaming decode state is single-utterance")
                past = state.pos
            cos_t, sin_t = self._rope_tables(past + N2)
            y = ws.get("tr.y", (B * N2, HS))
            qkv = ws.get("tr.qkv", (B * N2, 3 * HS))
            ao = ws.get("tr.ao", (B * N2, HS))
            hd = ws.get("tr.hd", (B * N2, int(mc.intermediate_size)))
            seg = dict(rows_per_seg=N2)
            new_kv: List[torch.Tensor] = []
            for li in range(int(mc.num_hidden_layers)):
                p = f"tr.{li}"
                self._ln_stream(X, y, w[p + ".ln1.w"], w[p + ".ln1.b"], B, N2, PADX, HS)
                hip.gemm(y, w[p + ".qkv.w"], qkv, M=B * N2, N=3 * HS, K=HS)
                hip.rope(qkv, cos_t, sin_t, rows=B * N2, rows_per_seg=N2, pos0=past, H=H, dh=dh, ldx=3 * HS)
                hip.rope(qkv, cos_t, sin_t, rows=B * N2, rows_per_seg=N2, pos0=past, H=H, dh=dh, ldx=3 * HS, x_off=HS)
                if state is None:
                    hip.attention(qkv, qkv, qkv, ao, B=B, H=H, dh=dh, Tq=N2, Tk=N2, ldq=3 * HS, ldk=3 * HS, ldv=3 * HS, ldo=HS,
                                  q_bstride=N2 * 3 * HS, k_bstride=N2 * 3 * HS, v_bstride=N2 * 3 * HS, o_bstride=N2 * HS,
                                  causal=True, window=win, k_off=HS, v_off=2 * HS)
                else:
                    # keys/values of earlier calls (post-RoPE) followed by this call's
                    cur = qkv[:, HS:].contiguous()  # [N2, 2*HS] = (k | v)
                    if state.kv is not None and state.kv_len > 0:
                        allkv = torch.cat([state.kv[li], cur], dim=0)
                    else:
                        allkv = cur
                    Tk = int(allkv.shape[0])
                    hip.attention(qkv, allkv, allkv, ao, B=1, H=H, dh=dh, Tq=N2, Tk=Tk, ldq=3 * HS, ldk=2 * HS, ldv=2 * HS, ldo=HS,
                                  q_bstride=0, k_bstride=0, v_bstride=0, o_bstride=0, causal=True, window=win, q_pos0=past,
                                  k_pos0=past + N2 - Tk, v_off=HS)
                    # DynamicSlidingWindowLayer keeps the last window-1 positions (installed transformers 5.x)
                    new_kv.append(allkv[-(win - 1):].clone())
                hip.gemm(ao, w[p + ".o.w"], X, M=B * N2, N=HS, K=HS, epilogue=hip.EPI_RES, R=X, scale=w[p + ".ls1"],
                         c_off=PADX * HS, r_off=PADX * HS, c_seg_stride=xs_stride, r_seg_stride=xs_stride, **seg)
                self._ln_stream(X, y, w[p + ".ln2.w"], w[p + ".ln2.b"], B, N2, PADX, HS)
                hip.gemm(y, w[p + ".fc1.w"], hd, M=B * N2, N=int(mc.intermediate_size), K=HS, epilogue=hip.EPI_GELU)
                hip.gemm(hd, w[p + ".fc2.w"], X, M=B * N2, N=HS, K=int(mc.intermediate_size), epilogue=hip.EPI_RES, R=X,
                         scale=w[p + ".ls2"], c_off=PADX * HS, r_off=PADX * HS, c_seg_stride=xs_stride, r_seg_stride=xs_stride, **seg)
            if state is not None:
                state.kv = new_kv
                state.kv_len = int(new_kv[0].shape[0])
                state.pos = past + N2
            # ---- SEANet decoder (HF:modeling_mimi.py:931-961)
            ch = int(mc.num_filters) * (2 ** len(mc.upsampling_ratios))  # 1024
            rows = N2
            # first conv k=7: window = 7 consecutive rows starting 6 rows before (the zero pad)
            Hc = ws.get("sea.h0", (B, 1 + rows, ch), zero=True)  # 1 zero row: x[t-1] of the transposed conv
            hip.gemm(X, w["sea.conv0.w"], Hc, M=B * rows, N=ch, K=int(mc.kernel_size) * HS, lda=HS, bias=w["sea.conv0.b"],
                     rows_per_seg=rows, a_seg_stride=xs_stride, c_off=ch, c_seg_stride=(1 + rows) * ch, ldc=ch)
            pad_in = 1
            for si, r in enumerate(mc.upsampling_ratios):
                r = int(r)
                co = ch // 2
                orow = rows * r
                Ho = ws.get(f"sea.h{si + 1}", (B, 2 + orow, co), zero=True)  # 2 zero rows: left pad of the k=3 conv
                # ELU -> ConvTranspose1d(ch -> co, k=2r, s=r): row t of A = [x[t-1] | x[t]]
                hip.gemm(Hc, w[f"sea.up{si}.w"], Ho, M=B * rows, N=r * co, K=2 * ch, lda=ch, bias=w[f"sea.up{si}.b"],
                         prologue=hip.PRO_ELU, rows_per_seg=rows, a_seg_stride=(pad_in + rows) * ch, a_off=(pad_in - 1) * ch,
                         c_off=2 * co, c_seg_stride=(2 + orow) * co, ldc=r * co)
                hid = co // int(mc.compress)
                last = si == len(mc.upsampling_ratios) - 1
                if last and self.fuse_tail and co == 64 and hid == 32 and int(mc.residual_kernel_size) == 3 and int(mc.last_kernel_size) == 3:
                    # last residual block + last conv in one kernel: the 64-channel 24 kHz activation is read once
                    wav = torch.empty(B, orow, device=dev)
                    hip.seanet_tail(Ho, w[f"sea.res{si}.c1.w"], w[f"sea.res{si}.c1.b"], w[f"sea.res{si}.c2.w"], w[f"sea.res{si}.c2.b"],
                                    w["sea.final.w"], self.final_bias, wav, B=B, T=orow, h_seg_stride=(2 + orow) * co, wav_seg_stride=orow)
                    Hc, ch, rows, pad_in = Ho, co, orow, 2
                    break
                # residual block: x + Conv1d(k=1)(ELU(Conv1d(k=3)(ELU(x))))
                Y1 = ws.get(f"sea.y{si + 1}", (B * orow, hid))
                hip.gemm(Ho, w[f"sea.res{si}.c1.w"], Y1, M=B * orow, N=hid, K=3 * co, lda=co, bias=w[f"sea.res{si}.c1.b"],
                         prologue=hip.PRO_ELU, rows_per_seg=orow, a_seg_stride=(2 + orow) * co)
                hip.gemm(Y1, w[f"sea.res{si}.c2.w"], Ho, M=B * orow, N=co, K=hid, bias=w[f"sea.res{si}.c2.b"], prologue=hip.PRO_ELU,
                         epilogue=hip.EPI_RES, R=Ho, rows_per_seg=orow, c_off=2 * co, r_off=2 * co, c_seg_stride=(2 + orow) * co,
                         r_seg_stride=(2 + orow) * co, ldc=co, ldr=co)
                Hc, ch, rows, pad_in = Ho, co, orow, 2
            else:
                wav = torch.empty(B, rows, device=dev)
                hip.final_conv(Hc, w["sea.final.w"], self.final_bias, wav, B=B, T=rows, h_seg_stride=(2 + rows) * ch, wav_seg_stride=rows)
        self.stream.synchronize()
        return wav

    def _ln_stream(self, X: torch.Tensor, y: torch.Tensor, wt: torch.Tensor, bs: torch.Tensor, B: int, N2: int, pad: int, HS: int) -> None:
        """LayerNorm of the zero-padded residual stream into a dense [B*N2, HS] buffer."""
        hip.norm(X, y, wt, rows=B * N2, C_=HS, eps=float(self.mc.norm_eps), kind=hip.NORM_LN, b=bs, rows_per_seg=N2,
                 x_off=pad * HS, x_seg_stride=(pad + N2) * HS)


class MimiStreamDecoder:
    """Chunked decode with a 2-frame token overlap and a growing transformer cache
    (reference: src/sopro/codec/mimi.py:83-181, as it behaves with the installed transformers 5.x:
    ``drop_cache_tail`` trims nothing)."""

    def __init__(self, codec: MimiCodec, overlap_frames: int = 2):
        self.codec = codec
        self.overlap_frames = int(overlap_frames)

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
        wav = self.codec.decode_batch(codes_in.unsqueeze(0), state=st)
        wav = wav[:, : (ov + n_new) * hop][:, ov * hop:]
        st.frames_seen += n_new
        st.samples_emitted += int(wav.shape[1])
        keep = min(self.overlap_frames, int(codes_in.shape[0]))
        st.tail_codes_tq = codes_in[-keep:].clone() if self.overlap_frames > 0 else None
        return wav, st
