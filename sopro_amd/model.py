"""Generation engine of the Sopro hot path on MI355X: the host-side mirror of the reference's
``SoproTTSModel`` (reference: src/sopro/model.py:53-401).

Same method names and argument meaning as the reference (``prepare_reference``,
``prepare_conditioning``, ``ar_stream``, ``nar_refine``, ``generate_tokens``) so that the CLI /
demo call sites keep working, plus batched forms (``*_batch``) that the reference does not have.
Every arithmetic step is a HIP kernel from ``libsopro_hip.so`` (see ``sopro_amd/hip.py``); torch is
used for device memory, index bookkeeping and streams only.  The per-frame autoregressive step is
recorded once into a hipGraph and replayed.
"""
from __future__ import annotations

import ctypes as C
import os

import math
from dataclasses import dataclass
from typing import Any, Dict, Iterator, List, Optional, Sequence, Tuple

import numpy as np
import torch

from . import hip
from .config import SoproTTSConfig
from .pack import pack_sopro, sinusoid_table

RMS_EPS = 1e-6  # reference: src/sopro/nn/blocks.py:27


@dataclass
class PreparedReference:
    """Field-for-field the reference's public type (src/sopro/model.py:45-50)."""

    ref_tokens_btq: torch.Tensor
    sv_ref: torch.Tensor
    ref_seq: torch.Tensor
    ref_kv_caches: List[Dict[str, Optional[torch.Tensor]]]


class Workspace:
    """Named device scratch buffers, reused across calls of the same shape (single stream)."""

    def __init__(self, device: torch.device):
        self.device = device
        self._bufs: Dict[Tuple[str, Tuple[int, ...], torch.dtype], torch.Tensor] = {}
        self.bytes = 0
        self.on_clear: List = []

    def get(self, name: str, shape: Sequence[int], dtype: torch.dtype = torch.float32, zero: bool = False) -> torch.Tensor:
        key = (name, tuple(int(s) for s in shape), dtype)
        t = self._bufs.get(key)
        if t is None:
            # buffers whose padding rows must read as zero are zero-filled once; kernels never write the pads
            t = (torch.zeros if zero else torch.empty)(key[1], dtype=dtype, device=self.device)
            self._bufs[key] = t
            self.bytes += t.numel() * t.element_size()
        return t

    def clear(self) -> None:
        for f in self.on_clear:  # recorded launch sequences that point into these buffers (every engine that shares them)
            f()
        self._bufs.clear()
        self.bytes = 0

    def over(self, budget_bytes: int) -> bool:
        """Buffers are kept per shape (recorded launch sequences point at them); a long-running process that sees many
        batch shapes calls this between batches and drops everything (with its recorded graphs) past the budget."""
        return self.bytes > budget_bytes


def _i32(values, device) -> torch.Tensor:
    return torch.tensor(list(values), dtype=torch.int32, device=device)


class SoproTTSModel:
    """Weights on the device + the orchestration of the hot path."""

    def __init__(self, cfg: SoproTTSConfig, weights: Dict[str, "np.ndarray"], device: str = "cuda:0", *, seed: int = 0,
                 use_graph: bool = True, precision: str = "f32"):
        hip.load()  # fail loudly here when the kernel library is missing
        if not torch.cuda.is_available():
            raise hip.SoproHipError("no HIP device visible: the Sopro engine has no CPU fallback")
        if precision not in ("f32", "bf16"):
            raise ValueError("precision must be 'f32' (parity with the fp32 reference) or 'bf16' (bf16 operands, fp32 accumulate)")
        # "bf16" = SURVEY.md 8d config 2: the NAR contractions round both operands to bf16 once (one MFMA pass instead of six)
        # and the AR frame streams bf16 weights (activations rounded to bf16 as MFMA operands); accumulators, norms, softmax,
        # the residual stream, ring buffers and the conditioning (text / reference encoders, cross-attention) stay fp32.
        self.precision = precision
        self.cfg = cfg
        self.device = torch.device(device)
        self.seed = int(seed)
        self.use_graph = bool(use_graph)
        self.D = int(cfg.d_model)
        self.V = int(cfg.codebook_size)
        self.Q = int(cfg.num_codebooks)
        packed = pack_sopro(weights, cfg)
        self.gates = {i: float(packed[f"ar.x_attns.{i}.gate_scale"][0]) for i in cfg.ar_xattn_layers}  # tanh(gate), text.py:131
        self._nar_mix = [(float(packed[f"nar.mix.{st}"][0]), float(packed[f"nar.mix.{st}"][1])) for st in cfg.stage_order()]  # nar.py:95-97
        self.w: Dict[str, torch.Tensor] = {k: v.to(self.device) for k, v in packed.items()}
        npos = int(cfg.pos_emb_max) + 8  # reference: src/sopro/model.py:62-64
        self.pe = sinusoid_table(npos, self.D).to(self.device)
        self.ws = Workspace(self.device)
        self._ref_ws = Workspace(self.device)  # prepare_reference's scratch (see there)
        self.stream = torch.cuda.Stream(device=self.device)
        self.bulk_stream = self.stream  # throughput-bound phase (NAR); a pipeline may point it at another CU partition
        self.prep_stream = self.stream
        self._driver = None  # the scheduler (PipelinedSynthesizer / ContinuousSynthesizer) that currently owns the streams
        self._ar_cache: Dict[Tuple[int, int, int], "_ARPlan"] = {}
        self._voice: Dict[int, Dict[str, Any]] = {}  # per-voice conditioning cache (see _voice_entry)
        self._voice_stacks: Dict[tuple, Any] = {}  # [U, Tr, D] K / V stacks per set of voices (see _voice_stack)
        self._film_stacks: Dict[tuple, Any] = {}  # [B, D] FiLM coefficient stacks per (set of voices, style strength)
        self._host_blocks: Dict[tuple, "hip.HostMirror"] = {}  # page-locked parameter blocks (see _host_block): an LRU
        self._recorded_blocks: Dict[tuple, "hip.HostMirror"] = {}  # ... and the ones recorded launch sequences hold the address of
        self._consts: Dict[tuple, torch.Tensor] = {}  # small constant int32 device vectors (see _const_i32)
        self._runs = [0]  # generation runs started so far (shared by the lanes of clone_lane): the sampler's default nonce
        self._nar_graphs = hip.GraphCache("nar_graph", cap=64)  # recorded NAR launch sequences per (B, T)
        self.ws_budget = int(os.environ.get("SOPRO_WS_BUDGET_GB", "16")) << 30  # scratch kept per batch shape, per engine
        # AR-step weights in the fragment order of the skinny kernel (1 KiB of consecutive memory per load instruction)
        self.wk: Dict[str, hip.SkinnyW] = {}
        with torch.cuda.device(self.device):
            for k, v in self.w.items():
                if (k.startswith("ar.blocks.") and k.endswith((".glu.w", ".ff1.w", ".ff2.w"))) or k == "ar.head.w" or \
                        (k.startswith("ar.x_attns.") and k.endswith((".qa.w", ".qu.w"))):
                    # bf16 mode: the frame's weight stream in bf16 (fp32 accumulate, fp32 norms / ring / residual)
                    self.wk[k] = hip.pack_skinny_w(v, glu=k.endswith(".glu.w"), bf16=(precision == "bf16"))
            torch.cuda.synchronize(self.device)
        # workgroup shapes of the AR-step stages, "<16-row groups>x<column tiles>" (sopro_skinny_args.mt / .nt; results do
        # not depend on them).  SOPRO_AR_TILES="glu:2x1,ff1:2x2,..." or "2x2" for all.
        self.ar_tiles = {"glu": "1x1", "ff1": "1x1", "ff2": "1x1", "head": "1x1"}
        self.set_ar_tiles(os.environ.get("SOPRO_AR_TILES", ""))
        # shape used instead for frames of more than 32 rows; a pipeline sets "1x2" on its 64-CU generation partition, where two
        # column tiles per workgroup are 5 % faster at 64 rows (profiles/r03_ar_tile_sweep_64rows.txt) and slower on the whole chip
        self.ar_tiles_wide: Optional[str] = None
        # the stage engine (csrc/stages.hip): the NAR launch sequence and its packed operands live in the library; the lanes of a
        # pipeline share it (read-only after finalize), each with its own workspace and stream
        from .stages import model_engine

        self.eng = model_engine(self)

    # ------------------------------------------------------------------ helpers
    def set_ar_tiles(self, spec: str) -> None:
        """'2x2' (every stage kind) or 'glu:2x1,ff1:2x2,ff2:2x2,head:1x2'.  Recorded frame graphs are dropped."""
        spec = (spec or "").strip()
        if not spec:
            return
        for part in spec.split(","):
            kind, _, shape = part.strip().rpartition(":")
            hip.ar_tile_code(shape)  # validates
            for k in ([kind] if kind else list(self.ar_tiles)):
                if k not in self.ar_tiles:
                    raise ValueError(f"unknown AR stage kind {k!r} (glu, ff1, ff2, head)")
                self.ar_tiles[k] = shape
        for plan in getattr(self, "_ar_cache", {}).values():
            plan.graph = None

    def on_stream(self, bulk: bool = False, prep: bool = False):
        """Context: run on the engine's stream, ordered after whatever the caller queued so far.  ``prep`` = the stream of the
        per-batch preparation (conditioning, text K/V folding): the AR stream itself unless a pipeline moved it to the
        throughput partition (these are GEMM-shaped and crawl on a small latency partition)."""
        s = self.bulk_stream if bulk else (self.prep_stream if prep else self.stream)
        s.wait_stream(torch.cuda.current_stream(self.device))
        return torch.cuda.stream(s)

    def clone_lane(self) -> "SoproTTSModel":
        """A second engine over the SAME device weights with its own streams, scratch and AR plans (pipelining)."""
        import copy

        other = copy.copy(self)
        other.ws = Workspace(self.device)
        other._ref_ws = Workspace(self.device)
        other.stream = torch.cuda.Stream(device=self.device)
        other.bulk_stream = other.stream
        other.prep_stream = other.stream
        other._driver = None
        other._ar_cache = {}
        other._voice = {}  # per-voice tensors are made on this lane's own preparation stream
        other._voice_stacks = {}
        other._film_stacks = {}
        other._host_blocks = {}  # a lane writes its blocks while another lane's kernels may still read theirs
        other._recorded_blocks = {}
        other._consts = {}
        other._nar_graphs = hip.GraphCache("nar_graph", cap=64)
        return other

    def _host_block(self, key: tuple, n: int, recorded: bool = False) -> "hip.HostMirror":
        """``n`` page-locked words that kernels of the library read or write (hip.HostMirror.copy_to / copy_from): how the
        host's small per-call parameters reach the device, and poll words the host, without a runtime copy on the path.  One block
        per use and shape; the user synchronises before it writes the block again.

        ``recorded=True``: the block's ADDRESS is held by recorded launch sequences (the refinement's lengths / range words).  Such
        blocks live in a dict of their own that is never evicted - a replayed sequence would otherwise read lengths from, and write
        its range word to, freed page-locked memory - and are dropped only together with the recorded sequences
        (``_drop_recorded``).  The others are an LRU of 64; an evicted block is freed only after this engine's streams have drained
        (a queued kernel may still read it), under the recording lock (hip.HostMirror.free)."""
        if recorded:
            hb = self._recorded_blocks.get(key)
            if hb is None or hb.n < int(n):
                if hb is not None:  # a recorded sequence holds the old address: drop the sequences with it
                    self._drop_recorded()
                hb = self._recorded_blocks[key] = hip.HostMirror(int(n))
            return hb
        hb = self._host_blocks.pop(key, None)
        if hb is not None and hb.n == int(n):  # (exact size: a key may carry id(plan), which a NEW plan of another batch size can inherit)
            self._host_blocks[key] = hb  # most recently used goes last
            return hb
        retired = [hb] if hb is not None else []
        while len(self._host_blocks) >= 64:
            retired.append(self._host_blocks.pop(next(iter(self._host_blocks))))
        if retired:
            for st in {id(x): x for x in (self.stream, self.bulk_stream, self.prep_stream)}.values():
                st.synchronize()
            for r in retired:
                r.free()
        hb = self._host_blocks[key] = hip.HostMirror(int(n))
        return hb

    def _drop_recorded(self) -> None:
        """Forget the recorded refinement sequences AND the page-locked blocks whose addresses they hold (in that order, after the
        stream that replays them has drained)."""
        self.bulk_stream.synchronize()
        self._nar_graphs.clear()
        for hb in self._recorded_blocks.values():
            hb.free()
        self._recorded_blocks.clear()

    def _const_i32(self, values: Sequence[int]) -> torch.Tensor:
        """A small constant int32 device vector (key lengths, voice indices of a batch), uploaded once per distinct content."""
        key = tuple(int(v) for v in values)
        t = self._consts.get(key)
        if t is None:
            if len(self._consts) >= 64:
                self._consts.pop(next(iter(self._consts)))
            t = self._consts[key] = _i32(list(key), self.device)
        return t

    def next_nonce(self, seed: Optional[int] = None) -> int:
        """Nonce of a generation run / slot admission, mixed into the sampler's Philox counter.  ``seed`` given: the run is
        reproducible (same seed, same text, same voice -> same audio).  ``None``: a process-wide run counter, so that
        every call is a new take - the reference draws from torch's global generator, which advances between calls
        (src/sopro/sampling.py:81-86)."""
        if seed is not None:
            return int(seed) & 0xFFFFFFFF
        self._runs[0] += 1
        return (0x9E3779B9 * self._runs[0]) & 0xFFFFFFFF

    def _voice_entry(self, ref: PreparedReference) -> Dict[str, Any]:
        """Everything conditioning needs from a voice that does not depend on the text, made once per PreparedReference
        object: dense [Tr, D] copies of the cached cross-attention K / V (the public type stores them
        [1, H, Tr, dh] like the reference, src/sopro/nn/ref.py:120-128) and, per style strength, the SpeakerFiLM
        coefficients.  Entries die with their PreparedReference (weakref)."""
        import weakref

        key = id(ref)
        ent = self._voice.get(key)
        if ent is not None and ent["ref"]() is ref:
            return ent
        dev, D = self.device, self.D
        kv = []
        for c in ref.ref_kv_caches:
            tu = int(c["k"].shape[2])
            kv.append((c["k"].to(dev).permute(0, 2, 1, 3).reshape(tu, D).contiguous(), c["v"].to(dev).permute(0, 2, 1, 3).reshape(tu, D).contiguous()))
        ent = {"ref": weakref.ref(ref, lambda _r, k=key, d=self._voice: d.pop(k, None)), "kv": kv, "film": {}}
        self._voice[key] = ent
        return ent

    def _voice_stack(self, order: Sequence[PreparedReference], Tr: int) -> List[Tuple[torch.Tensor, torch.Tensor]]:
        """Dense [U, Tr, D] K / V stacks of a set of voices for the reference cross-attention layers, kept while the same
        PreparedReference objects come back (a service batches the same speakers again and again; bench.py's 32 voices)."""
        import weakref

        key = tuple(id(r) for r in order) + (int(Tr),)
        ent = self._voice_stacks.get(key)
        if ent is not None and all(wr() is r for wr, r in zip(ent[0], order)):
            return ent[1]
        dev, D = self.device, self.D
        out = []
        for i in range(int(self.cfg.ref_xattn_layers)):
            Ku, Vu = torch.zeros(len(order), Tr, D, device=dev), torch.zeros(len(order), Tr, D, device=dev)
            for u, r in enumerate(order):
                ku, vu = self._voice_entry(r)["kv"][i]
                tu = int(ku.shape[0])
                Ku[u, :tu], Vu[u, :tu] = ku, vu
            out.append((Ku, Vu))
        if len(self._voice_stacks) >= 8:
            self._voice_stacks.clear()
        self._voice_stacks[key] = ([weakref.ref(r) for r in order], out)
        return out

    def rf_ar(self) -> int:
        return self.cfg.rf_ar()

    def rf_nar(self) -> int:
        return self.cfg.rf_nar()

    # ------------------------------------------------------------------ per-voice preparation
    @torch.inference_mode()
    def prepare_reference(self, ref_tokens_tq: torch.Tensor) -> PreparedReference:
        """reference: src/sopro/model.py:151-170 (Token2SV src/sopro/nn/speaker.py:37-61, reference encoder model.py:133-149,
        K/V caches src/sopro/nn/ref.py:120-128).  The launch sequence is ``sopro_ref_prepare`` (csrc/stages.hip)."""
        cfg, dev, D = self.cfg, self.device, self.D
        if ref_tokens_tq.dim() != 2 or ref_tokens_tq.shape[1] != self.Q:
            raise ValueError(f"ref_tokens_tq must be [T, {self.Q}], got {tuple(ref_tokens_tq.shape)}")
        T, L, H = int(ref_tokens_tq.shape[0]), int(cfg.ref_xattn_layers), int(cfg.ref_xattn_heads)
        with self.on_stream():
            tok64 = ref_tokens_tq.to(dev).long()
            tok = tok64.to(torch.int32).contiguous()
            sv = torch.empty(1, int(cfg.sv_student_dim), device=dev)
            ref_seq = torch.empty(1, T, D, device=dev)
            kvs = [torch.empty(T, 2 * D, device=dev) for _ in range(L)]
            # own scratch: a client thread may prepare a voice while a scheduler drives this engine's other streams
            wsb = self._ref_ws.get("ref.ws", (int(hip.load().sopro_ref_workspace_bytes(self.eng.h, T)) // 4 + 64,))
            hip.ref_prepare(self.eng.h, wsb, tok, T, sv, ref_seq, kvs)
            caches: List[Dict[str, Optional[torch.Tensor]]] = []
            for kv in kvs:  # [1, H, T, dh] views, as the reference stores them
                caches.append({"k": kv[:, :D].reshape(1, T, H, D // H).permute(0, 2, 1, 3), "v": kv[:, D:].reshape(1, T, H, D // H).permute(0, 2, 1, 3),
                               "key_padding_mask": None})
        self.stream.synchronize()
        return PreparedReference(ref_tokens_btq=tok64.unsqueeze(0), sv_ref=sv, ref_seq=ref_seq, ref_kv_caches=caches)

    @torch.inference_mode()
    def token2sv(self, ref_btq: torch.Tensor, lengths: Optional[torch.Tensor] = None) -> torch.Tensor:
        """Speaker vector of one voice's codec tokens, [1, T, Q] -> [1, sv_student_dim] (reference: the ``token2sv`` module,
        src/sopro/nn/speaker.py:37-61, as ``SoproTTS.encode_speaker`` calls it, src/sopro/model.py:457-475).  The launch
        sequence is the Token2SV part of ``sopro_ref_prepare`` (csrc/stages.hip)."""
        if ref_btq.dim() != 3 or ref_btq.shape[0] != 1 or ref_btq.shape[2] != self.Q:
            raise ValueError(f"ref_btq must be [1, T, {self.Q}], got {tuple(ref_btq.shape)}")
        T = int(ref_btq.shape[1])
        if lengths is not None and int(lengths.reshape(-1)[0]) != T:
            raise ValueError("lengths must equal the token count (one unpadded voice)")
        with self.on_stream():
            tok = ref_btq[0].to(self.device).to(torch.int32).contiguous()
            sv = torch.empty(1, int(self.cfg.sv_student_dim), device=self.device)
            wsb = self._ref_ws.get("ref.ws", (int(hip.load().sopro_ref_workspace_bytes(self.eng.h, T)) // 4 + 64,))
            hip.ref_prepare(self.eng.h, wsb, tok, T, sv, None, None)
        self.stream.synchronize()
        return sv

    # ------------------------------------------------------------------ per-utterance conditioning
    @torch.inference_mode()
    def prepare_conditioning(self, text_ids_1d: torch.Tensor, ref: PreparedReference, *, max_frames: int,
                             device: Any = None, style_strength: float = 1.0) -> Dict[str, torch.Tensor]:
        """reference: src/sopro/model.py:172-216 (one utterance)."""
        out = self.prepare_conditioning_batch([text_ids_1d], [ref], max_frames=max_frames, style_strength=style_strength)
        S = int(out["text_lens_host"][0])
        return {"txt_seq": out["txt_seq"][:, :S], "text_mask": torch.ones(1, S, dtype=torch.bool, device=self.device),
                "txt_pool": out["txt_pool"], "sv_ref": ref.sv_ref, "cond_ar": out["cond_ar"]}

    @torch.inference_mode()
    def prepare_conditioning_batch(self, ids_list: Sequence[torch.Tensor], refs: Sequence[PreparedReference], *,
                                   max_frames: int, style_strength: float = 1.0, cond_out: Optional[torch.Tensor] = None) -> Dict[str, Any]:
        """B utterances at once (new): text encoder (src/sopro/nn/text.py:29-44), base = pooled text + frame positions
        (model.py:200-202), SpeakerFiLM (src/sopro/nn/speaker.py:76-85), three reference cross-attention blocks
        (src/sopro/nn/ref.py:54-108), cond_norm (model.py:208).  The launch sequence is ``sopro_cond_prepare`` (csrc/stages.hip);
        this method pads the ids, keeps what depends on the voices only (FiLM coefficients, dense K / V) and marshals."""
        cfg, dev, D = self.cfg, self.device, self.D
        B = len(ids_list)
        if B == 0 or len(refs) != B:
            raise ValueError("need one reference per utterance")
        lens_h = [int(x.numel()) for x in ids_list]
        S = max(lens_h)
        if S > int(cfg.max_text_len) or min(lens_h) <= 0:
            raise ValueError(f"text length must be in [1, {cfg.max_text_len}]")
        Tar = int(max_frames) + 1
        if Tar > self.pe.shape[0]:
            raise ValueError("max_frames exceeds the position table")
        lib, eng = hip.load(), self.eng
        with self.on_stream(prep=True):
            # ids and lengths travel through one page-locked block and a kernel of the library (no runtime copy on the path;
            # the block is rewritten by the next call of this engine, which comes after this call's synchronize below)
            # (keyed on a rounded capacity: a serving process sees every text length, and one block per (B, S) would walk through
            # the LRU in minutes)
            S_cap = (S + 31) // 32 * 32
            hb = self._host_block(("cond.in", B, S_cap), B * S_cap + B)
            arr = hb.array()[: B * S + B]
            arr[: B * S] = 0
            for b, x in enumerate(ids_list):
                arr[b * S: b * S + lens_h[b]] = x.detach().to("cpu", torch.int32).view(-1).numpy()
            arr[B * S:] = lens_h
            blk = torch.empty(B * S + B, dtype=torch.int32, device=dev)
            hb.copy_to(blk)
            ids, lens = blk[: B * S].view(B, S), blk[B * S:]
            s = float(style_strength)
            # SpeakerFiLM coefficients depend on the voice (and the style strength) only: computed once per voice and kept
            # with its dense K / V (self._voice), so a call with known voices launches nothing for them  (speaker.py:76-85)
            vcs = [self._voice_entry(r) for r in refs]
            todo = [i for i, vc in enumerate(vcs) if s not in vc["film"]]
            if todo:
                uniq = list({id(vcs[i]): i for i in todo}.values())
                n = len(uniq)
                sv_u = torch.cat([refs[i].sv_ref.to(dev).reshape(1, -1) for i in uniq], dim=0).contiguous()
                mul_u, add_u = torch.empty(n, D, device=dev), torch.empty(n, D, device=dev)
                hip.film_coeffs(eng.h, sv_u, s, n, torch.empty(n * 5 * D, device=dev), mul_u, add_u)
                for j, i in enumerate(uniq):
                    vcs[i]["film"][s] = (mul_u[j], add_u[j])
            # ... and so do their [B, D] stacks for a SET of voices (a serving process sees the same sets again and again: one
            # `cat` the first time, nothing afterwards)
            skey = (tuple(id(vc) for vc in vcs), s)
            st = self._film_stacks.get(skey)
            if st is None:
                if all(vc is vcs[0] for vc in vcs):
                    mul, add = (t.unsqueeze(0).expand(B, D).contiguous() for t in vcs[0]["film"][s])
                else:
                    mul = torch.stack([vc["film"][s][0] for vc in vcs]).contiguous()
                    add = torch.stack([vc["film"][s][1] for vc in vcs]).contiguous()
                sv = torch.cat([r.sv_ref.to(dev).reshape(1, -1) for r in refs], dim=0).contiguous()
                if len(self._film_stacks) >= 16:
                    self._film_stacks.pop(next(iter(self._film_stacks)))
                st = self._film_stacks[skey] = (mul, add, sv, list(vcs))  # (the entries keep the ids alive)
            mul, add, sv = st[0], st[1], st[2]
            tr_h = [int(r.ref_kv_caches[0]["k"].shape[2]) for r in refs]
            Tr = max(tr_h)
            klens = self._const_i32(tr_h) if min(tr_h) != Tr else None
            # rows that share a voice (the same PreparedReference object) share its cached K / V: one dense copy per voice and
            # layer, read through a zero batch stride (one voice), the rows' own blocks (B voices) or an index per row
            order: List[PreparedReference] = []
            seen: Dict[int, int] = {}
            for r in refs:
                if id(r) not in seen:
                    seen[id(r)] = len(order)
                    order.append(r)
            U = len(order)
            kv_index = None if U in (1, B) else self._const_i32([seen[id(r)] for r in refs])
            if U == 1:  # the voice's own dense [Tr, D] copies, made once (self._voice)
                kvs = self._voice_entry(order[0])["kv"]
            else:  # one [U, Tr, D] stack per layer, made once per set of voices
                kvs = self._voice_stack(order, Tr)
            txt_seq, txt_pool = torch.empty(B, S, D, device=dev), torch.empty(B, D, device=dev)
            # (a scheduler hands over the AR plan's own conditioning buffer: the AR loop and the refinement then read it in place)
            cond_ar = cond_out if cond_out is not None else torch.empty(B, Tar, D, device=dev)
            if tuple(cond_ar.shape) != (B, Tar, D) or not cond_ar.is_contiguous():
                raise ValueError("cond_out must be a contiguous [B, max_frames + 1, D] tensor")
            wsb = self.ws.get("cond.ws", (int(lib.sopro_cond_workspace_bytes(eng.h, B, S, Tar)) // 4 + 64,))
            hip.cond_prepare(eng.h, wsb, ids, lens, min(lens_h) != S, mul, add, [k for k, _v in kvs], [v for _k, v in kvs],
                             0 if (U == 1 and B > 1) else Tr * D, kv_index, klens, B, S, Tar, Tr, txt_seq, txt_pool, cond_ar)
        self.prep_stream.synchronize()
        return {"txt_seq": txt_seq, "text_lens": lens, "text_lens_host": lens_h, "txt_pool": txt_pool, "sv_ref": sv,
                "cond_ar": cond_ar}

    # ------------------------------------------------------------------ autoregressive generation
    def _ar_plan(self, B: int, S_cap: int, Tar: int) -> "_ARPlan":
        """The cached plan of a shape, or - while a suspended ar_stream / stream() generator of the same shape still owns
        that one - a private plan for this run (the reference keeps its state local to each call, model.py:218-305)."""
        key = (B, S_cap, Tar)
        plan = self._ar_cache.get(key)
        if plan is None:
            if len(self._ar_cache) >= 8:
                self._ar_cache.clear()  # plans still in use stay alive through their runs
            plan = _ARPlan(self, B, S_cap, Tar)
            self._ar_cache[key] = plan
        elif plan.owner is not None and plan.owner() is not None and not plan.owner().done:
            plan = _ARPlan(self, B, S_cap, Tar)  # not cached: dropped with its run
        return plan

    @torch.inference_mode()
    def ar_generate_batch(self, cond_ar: torch.Tensor, txt_seq: torch.Tensor, text_lens: Optional[torch.Tensor], *,
                          max_frames: int, top_p: float = 0.9, temperature: float = 1.05, anti_loop: bool = True,
                          min_gen_frames: Optional[int] = None, stop_on_first_eos: bool = False,
                          poll_every: int = 16, seed: Optional[int] = None, run: Optional["_ARRun"] = None,
                          inplace: bool = False) -> Tuple[torch.Tensor, List[int]]:
        """Run the AR loop for B rows until every row has stopped or max_frames+1 steps were taken
        (reference loop: src/sopro/model.py:218-305, one row).  Returns (hist [B, steps] int32 on the
        device, per-row frame counts T_b following generate_tokens' cut at the FIRST EOS, model.py:385-390)."""
        B, Tar, _ = cond_ar.shape
        if Tar != int(max_frames) + 1:
            raise ValueError("cond_ar must have max_frames+1 rows")
        if run is None:
            run = _ARRun(self, cond_ar, txt_seq, text_lens, top_p=top_p, temperature=temperature, anti_loop=anti_loop,
                         min_gen_frames=min_gen_frames, seed=seed)
        # The stop poll trails the launches by one chunk: chunk k+1 is enqueued before the host looks at chunk k's counter, so
        # the GPU never waits for the host round trip (rows that have stopped are masked on the device; the extra frames of
        # a batch that turns out to be finished are discarded below).
        poll_every = int(hip.dev_env("SOPRO_AR_POLL", str(poll_every)))
        steps = 0
        pending = None
        ev0 = None
        if hip.phase_log is not None:
            ev0 = torch.cuda.Event(enable_timing=True)
            ev0.record(self.stream)
        # While no row of a batch has stopped yet, frames are enqueued four polls' worth at a time: the launch thread then
        # has ~15 ms of queued work ahead of the GPU instead of ~4, which is what it takes to ride out a descheduled host
        # thread on a busy machine (measured: 14.5 k vs 17.6 k audio-s/s on the same build, load average 16).  Once the first
        # row has stopped - the batch may be over any frame now - and for small batches (latency matters, all rows tend to stop
        # together) the poll is back to every `poll_every` frames.
        seen_stop = B < 8
        while steps < Tar:
            n = min(int(poll_every) * (1 if seen_stop else 4), Tar - steps)
            run.advance(n)
            steps += n
            if pending is not None:
                got = pending()
                seen_stop = seen_stop or got > 0
                if got >= B:
                    break
            pending = run.poll_async(stop_on_first_eos)
        if ev0 is not None:
            ev1 = torch.cuda.Event(enable_timing=True)
            ev1.record(self.stream)
            hip.phase_log.append((steps, B, ev0, ev1))
        hist, first_eos = run.history(steps, clone=not inplace)  # (``inplace``: a view of the plan's history, see _ARRun.history)
        run.done = True
        lens = [int(f) if f >= 0 else steps for f in first_eos]
        return hist, lens

    @torch.inference_mode()
    def ar_stream(self, prep: Dict[str, torch.Tensor], *, max_frames: int = 400, top_p: float = 0.9,
                  temperature: float = 1.05, anti_loop: bool = True, use_prefix: bool = False,
                  prefix_sec_fixed: Optional[float] = None, use_stop_head: Optional[bool] = None,
                  stop_patience: Optional[int] = None, stop_threshold: Optional[float] = None,
                  min_gen_frames: Optional[int] = None, lookahead: int = 1, seed: Optional[int] = None
                  ) -> Iterator[Tuple[int, int, bool]]:
        """Yields (t, token, is_eos) like the reference generator (src/sopro/model.py:218-305).
        ``lookahead`` frames are generated per host round trip (new; 1 == token-by-token); ``seed`` (new) pins the
        sampler's draws for this run."""
        cond = prep["cond_ar"]
        Tar = int(max_frames) + 1
        if cond.shape[1] != Tar:
            raise ValueError("prep['cond_ar'] must have max_frames+1 rows")
        min_gen = int(min_gen_frames if min_gen_frames is not None else self.cfg.min_gen_frames)
        run = _ARRun(self, cond, prep["txt_seq"], None, top_p=top_p, temperature=temperature, anti_loop=anti_loop,
                     min_gen_frames=min_gen, seed=seed)
        t = 0
        eos = self.V
        try:
            while t < Tar:
                n = min(max(1, int(lookahead)), Tar - t)
                run.advance(n)
                toks = run.tokens_host(t, t + n)
                for j in range(n):
                    tok = int(toks[j])
                    is_eos = tok == eos
                    yield t + j, tok, is_eos
                    if is_eos and (t + j + 1) >= min_gen:
                        return
                t += n
        finally:
            run.done = True  # also when the consumer abandons the generator: the plan is free again

    # ------------------------------------------------------------------ NAR refinement
    @torch.inference_mode()
    def nar_refine(self, cond_seq: torch.Tensor, tokens_A_1xT: torch.Tensor, lens: Optional[Sequence[int]] = None, *, sync: bool = True,
                   raw: bool = False) -> torch.Tensor:
        """Codebooks 1..Q-1 from codebook 0: [B, T, D], [B, T] -> [B, T, Q] int64
        (reference: src/sopro/model.py:307-347 and src/sopro/nn/nar.py:89-116; B = 1 there).  The launch sequence is
        ``sopro_nar_refine_io`` (csrc/stages.hip); this method stages the inputs and records / replays it per (B, T) shape."""
        dev, D = self.device, self.D
        B, T, _ = cond_seq.shape
        lens_l = [T] * B if lens is None else [int(n) for n in lens]
        self._nar_make_room(B, T)
        with self.on_stream(bulk=True):
            # inputs land in persistent buffers so that the launch sequence of a (B, T) shape can be recorded once
            cond = self.ws.get("nar.cond", (B * T, D))
            cond.view(B, T, D).copy_(cond_seq.to(dev).float())
            rvq1 = self.ws.get("nar.rvq1", (B, T), dtype=torch.int32)
            rvq1.copy_(tokens_A_1xT.to(dev).reshape(B, T).to(torch.int32))
        return self._nar_pass(cond, T * D, rvq1, T, lens_l, B, T, sync=sync, raw=raw)

    def _nar_make_room(self, B: int, T: int) -> None:
        if self.ws.over(self.ws_budget) and not any(k[:2] == (B, T) for k in self._nar_graphs.graphs):
            torch.cuda.synchronize(self.device)
            self._nar_graphs.clear()
            self.ws.clear()

    @torch.inference_mode()
    def _nar_pass(self, cond: torch.Tensor, cond_bstride: int, cb0: torch.Tensor, cb0_bstride: int, lens_l: Sequence[int], B: int, T: int, *,
                  sync: bool, raw: bool, safe: bool = False) -> torch.Tensor:
        """One refinement pass on operands that stay where they are (``cond``: [B, >= T rows, D] blocks ``cond_bstride`` floats apart,
        ``cb0``: codebook 0 as rows of a history buffer ``cb0_bstride`` ints apart) - the AR plan's own buffers when a scheduler
        calls (phase_nar), this engine's staging buffers for the public method.  The lengths reach the device through a page-locked
        block that the recorded sequence itself reads; the pass's RANGE EVENTS (f16 operands that had to be saturated, see
        sopro_gemm_split_ext.range_events) come back in another one, checked by ``nar_guard`` once the stream has been synchronised."""
        Q = self.Q
        lib, eng = hip.load(), self.eng
        with self.on_stream(bulk=True):
            hl = self._host_block(("nar.lens", B), B, recorded=True)
            hl.array()[:B] = [int(n) for n in lens_l]
            hr = self._host_block(("nar.range",), 1, recorded=True)
            toks = self.ws.get("nar.toks", (B * T, Q), dtype=torch.int32)
            scratch = self.ws.get(f"nar.stage_ws.{B}x{T}", (int(lib.sopro_nar_workspace_bytes(eng.h, B, T)),), dtype=torch.uint8)
            io = hip.NarIO(cond.data_ptr(), int(cond_bstride), cb0.data_ptr(), int(cb0_bstride), hl.ptr, toks.data_ptr(), hr.ptr, 1 if safe else 0)

            def issue():
                hip._check(lib.sopro_nar_refine_io(eng.h, scratch.data_ptr(), C.byref(io), B, T, hip._stream()), "sopro_nar_refine_io")

            if self.use_graph and os.environ.get("SOPRO_NO_BULK_GRAPH", "0") != "1":
                # (the recorded sequence holds the operands' addresses: they are part of its key)
                self._nar_graphs.run((B, T, cond.data_ptr(), int(cond_bstride), cb0.data_ptr(), int(cb0_bstride), bool(safe), hl.ptr, hr.ptr), issue)
            else:
                issue()
            self._nar_last = (cond, cond_bstride, cb0, cb0_bstride, list(lens_l), B, T)
            # ``raw`` (a scheduler that decodes on the same stream right away): the engine's own int32 buffer, no widening copy
            out = toks.view(B, T, Q) if raw else None
        if sync:  # (``sync=False``: the caller stays on the bulk stream - e.g. decodes there - synchronises once, later, and calls nar_guard)
            self.bulk_stream.synchronize()
            if self.nar_guard(redo=True):
                self.bulk_stream.synchronize()
        if out is None:
            with self.on_stream(bulk=True):
                out = toks.view(B, T, Q).long()
            if sync:
                self.bulk_stream.synchronize()
        return out

    def nar_guard(self, redo: bool = True) -> bool:
        """After the stream of the last refinement pass has been synchronised: did that pass leave the f16 operands' range?  If so
        (and ``redo``) the same pass is queued again on the six-pass bf16 operands - fp32's exponent range - into the same token
        buffer, and True is returned: the caller repeats whatever consumed the tokens (a checkpoint whose residual
        stream exceeds fp16 pays the slower path instead of getting wrong tokens silently)."""
        hr = self._recorded_blocks.get(("nar.range",))
        if hr is None or self.precision != "f32" or int(hr.values()[0]) == 0:
            return False
        self.range_fallbacks = getattr(self, "range_fallbacks", 0) + 1
        if redo:
            cond, cbs, cb0, c0s, lens_l, B, T = self._nar_last
            hr.array()[0] = 0
            self._nar_pass(cond, cbs, cb0, c0s, lens_l, B, T, sync=False, raw=True, safe=True)
        return True

    # ------------------------------------------------------------------ text + reference -> tokens
    @torch.inference_mode()
    def generate_tokens(self, text_ids: torch.Tensor, ref: PreparedReference, *, max_frames: int, device: Any = None,
                        top_p: float = 0.9, temperature: float = 1.05, anti_loop: bool = True, use_prefix: bool = False,
                        prefix_sec_fixed: Optional[float] = None, style_strength: float = 1.0,
                        use_stop_head: Optional[bool] = None, stop_patience: Optional[int] = None,
                        stop_threshold: Optional[float] = None, min_gen_frames: Optional[int] = None,
                        seed: Optional[int] = None) -> torch.Tensor:
        """reference: src/sopro/model.py:349-401 -> [T, Q] int64 (``[0, Q]`` when EOS comes first)."""
        return self.generate_tokens_batch([text_ids], [ref], max_frames=max_frames, top_p=top_p, temperature=temperature,
                                          anti_loop=anti_loop, style_strength=style_strength, min_gen_frames=min_gen_frames,
                                          seed=seed)[0]

    @torch.inference_mode()
    def generate_tokens_batch(self, ids_list: Sequence[torch.Tensor], refs: Sequence[PreparedReference], *, max_frames: int,
                              top_p: float = 0.9, temperature: float = 1.05, anti_loop: bool = True,
                              style_strength: float = 1.0, min_gen_frames: Optional[int] = None,
                              timings: Optional[Dict[str, float]] = None, seed: Optional[int] = None) -> List[torch.Tensor]:
        """B utterances -> list of [T_b, Q] int64 token matrices (new, batched form of generate_tokens)."""
        ev = _PhaseTimer(self.stream, timings)
        state = self.phase_ar(ids_list, refs, max_frames=max_frames, top_p=top_p, temperature=temperature, anti_loop=anti_loop,
                              style_strength=style_strength, min_gen_frames=min_gen_frames, ev=ev, seed=seed)
        toks = self.phase_nar(state)
        ev.mark("nar")
        return toks

    def plan_for(self, ids_list, max_frames: int) -> "_ARPlan":
        """The AR plan a batch of these texts will generate on (a scheduler asks for it BEFORE the conditioning stage, which then
        writes ``cond_ar`` straight into the plan's buffer; the refinement later reads that buffer and the plan's token history in
        place: no copy between the stages of a pass)."""
        S = max(int(x.numel()) for x in ids_list)
        return self._ar_plan(len(ids_list), ((S + 63) // 64) * 64, int(max_frames) + 1)

    def phase_cond(self, ids_list, refs, *, max_frames, style_strength, ev=None, plan=None):
        """Per-batch conditioning (GEMM-shaped; runs on the preparation stream, needs no generation slot)."""
        prep = self.prepare_conditioning_batch(ids_list, refs, max_frames=max_frames, style_strength=style_strength,
                                               cond_out=plan.cond if plan is not None else None)
        prep["plan"] = plan
        if ev is not None:
            ev.mark("cond")
        return prep

    @torch.inference_mode()
    def ar_prepare(self, prep, *, top_p, temperature, anti_loop, min_gen_frames, seed=None, nonces=None, row_ids=None) -> "_ARRun":
        """The per-batch preparation of the AR phase (plan buffers, folded text operands of the cross-attention layers: ~30
        GEMM-shaped launches on the preparation stream).  Needs no generation slot: a scheduler calls it while the batch waits
        for one and hands the run to phase_ar, so the slot only ever replays frames."""
        return _ARRun(self, prep["cond_ar"], prep["txt_seq"], prep["text_lens"], top_p=top_p, temperature=temperature,
                      anti_loop=anti_loop, min_gen_frames=min_gen_frames, seed=seed, nonces=nonces, row_ids=row_ids,
                      text_lens_host=prep.get("text_lens_host"), plan=prep.get("plan"))

    def phase_ar(self, ids_list, refs, *, max_frames, top_p, temperature, anti_loop, style_strength, min_gen_frames, ev=None,
                 prep=None, seed=None, run=None):
        """Latency-bound half of generate_tokens_batch: (conditioning +) the AR graph replay."""
        if prep is None:
            prep = self.phase_cond(ids_list, refs, max_frames=max_frames, style_strength=style_strength, ev=ev)
        # a prepared run on a plan that also holds the conditioning (a scheduler's pass): the history stays in the plan as well
        inplace = run is not None and prep.get("plan") is not None and run.plan is prep["plan"]
        hist, lens = self.ar_generate_batch(prep["cond_ar"], prep["txt_seq"], prep["text_lens"], max_frames=max_frames,
                                            top_p=top_p, temperature=temperature, anti_loop=anti_loop,
                                            min_gen_frames=min_gen_frames, seed=seed, run=run, inplace=inplace)
        if ev is not None:
            ev.mark("ar")
        return {"cond_ar": prep["cond_ar"], "hist": hist, "lens": lens, "B": len(ids_list), "plan": run.plan if inplace else None}

    def phase_nar(self, state, full: bool = False, sync: bool = True, raw: bool = False):
        """Throughput-bound half: NAR refinement of the generated codebook-0 tokens -> one [T_b, Q] matrix per utterance, or with
        ``full`` the whole padded [B, Tn, Q] batch (rows hold valid but meaningless codes past their own length)."""
        lens, hist, B = state["lens"], state["hist"], state["B"]
        Tm = max(lens)
        if Tm <= 0:
            if full:
                return torch.zeros(B, 0, self.Q, dtype=torch.long, device=self.device)
            return [torch.zeros(0, self.Q, dtype=torch.long, device=self.device) for _ in range(B)]
        # a few frames of padding keep the set of batch shapes (scratch + recorded graphs per shape) small
        Tm = min(-(-Tm // 8) * 8, int(hist.shape[1]), int(state["cond_ar"].shape[1]))
        plan = state.get("plan")
        if plan is not None:
            # the plan's conditioning rows and token history are read where they are (EOS codes of stopped rows are clamped by the
            # stage's own seeding kernel; rows past their own length are ignored below)
            self._nar_make_room(B, Tm)
            toks = self._nar_pass(plan.cond, plan.Tar * self.D, plan.hist, int(plan.hist.shape[1]), [max(1, n) for n in lens], B, Tm,
                                  sync=sync, raw=raw)
        else:
            rvq1 = hist[:, :Tm].clamp(max=self.V - 1)  # rows past their own length are ignored below
            toks = self.nar_refine(state["cond_ar"][:, :Tm, :], rvq1, lens=[max(1, n) for n in lens], sync=sync, raw=raw)
        if full:
            return toks
        return [toks[b, : lens[b]] for b in range(B)]


class _PhaseTimer:
    """Optional per-phase wall-clock (device-synchronised) used by bench.py."""

    def __init__(self, stream: torch.cuda.Stream, sink: Optional[Dict[str, float]]):
        self.sink = sink
        self.stream = stream
        if sink is not None:
            import time

            self._time = time
            stream.synchronize()
            self.t0 = time.perf_counter()

    def mark(self, name: str) -> None:
        if self.sink is None:
            return
        self.stream.synchronize()
        t = self._time.perf_counter()
        self.sink[name] = self.sink.get(name, 0.0) + (t - self.t0)
        self.t0 = t


class _ARPlan:
    """Static buffers + the recorded per-frame launch sequence for (B rows, S_cap keys, Tar frames).
    reference step: src/sopro/nn/generator.py:98-130."""

    def __init__(self, m: SoproTTSModel, B: int, S_cap: int, Tar: int, slots: bool = False):
        cfg, dev, D = m.cfg, m.device, m.D
        self.m, self.B, self.S_cap, self.Tar = m, B, S_cap, Tar
        self.max_steps = Tar
        V1 = m.V + 1
        z = lambda *s, dt=torch.float32: torch.zeros(*s, dtype=dt, device=dev)  # noqa: E731
        # the small per-run parameters live in ONE device block, filled from one page-locked host block by one kernel of the
        # library (_ARRun): sampling parameters (8 floats) | Philox key (seed: one recorded frame serves every seed) | per-row run
        # nonce of the sampler | the row's identity in the sampler's counter (see _ARRun) | text lengths
        self.pblock = z(10 + 3 * B, dt=torch.int32)
        self.params = self.pblock[0:8].view(torch.float32)
        self.key = self.pblock[8:10]
        self.nonce = self.pblock[10:10 + B]
        self.row_id = self.pblock[10 + B:10 + 2 * B]
        self.row_id.copy_(torch.arange(B, dtype=torch.int32, device=dev))
        self.cond = z(B, Tar, D)
        self.x = [z(B, D) for _ in range(4)]
        self.part = z(4 * D // 384, B, D)
        self.u = z(B, 4 * D)
        self.q = z(B, D)
        self.att = z(B, D)
        self.logits = z(B, V1)
        # folded cross-attention operands per layer: K' = K_h Wq_h, V' = V_h Wo_h^T, [B, H, S_cap, D].  bf16 mode keeps what the
        # frame streams EVERY frame - these operands and the ring buffers - as bf16 in memory (store_format 1 of sopro_ar_frame):
        # half the bytes of the frame's two largest state streams; they are folded in fp32 (one scratch pair) and rounded once.
        self.bf16_state = m.precision == "bf16" and hip.dev_env("SOPRO_BF16_STATE", "1") != "0"
        sdt = torch.bfloat16 if self.bf16_state else torch.float32
        # Unfolded keys (round 4, sopro_ar_frame.k_unfold): kp holds K [B, S_cap, D] instead of the folded K' [B, 4, S_cap, D] - a quarter
        # of the key bytes the frame streams - and the query rides on the feed-forward launches (pack.py "qa.w" / "qu.w" / "q.b").
        # Slot plans (continuous batching) keep the folded form: their per-row admission copies are written for it.
        # Measured (profiles/r04_experiments.md): fp32 frame 380 -> 357 us per 64 rows in the pipeline, tokens exact on every fixture;
        # with bf16 operands the keys are small already and the extra feed-forward columns cost what they save (24.3 -> 25.4 ms per
        # step of AR phases) - the bf16 mode keeps the folded form.
        self.k_unfold = (not slots) and hip.dev_env("SOPRO_AR_KUNFOLD", "1" if m.precision == "f32" else "0") != "0"
        if self.k_unfold:
            self.kp = {i: z(B, S_cap, D, dt=sdt) for i in cfg.ar_xattn_layers}
            self.qa, self.qpart = z(B, D), z(4 * D // 384, B, D)
        else:
            self.kp = {i: z(B, 4, S_cap, D, dt=sdt) for i in cfg.ar_xattn_layers}
        self.vp = {i: z(B, 4, S_cap, D, dt=sdt) for i in cfg.ar_xattn_layers}
        self.fold32 = (z(B, 4, S_cap, D), z(B, 4, S_cap, D)) if self.bf16_state else None  # (unfolded keys use the first quarter of [0])
        self.xp = z(4, B, D)  # per-head partial outputs of the cross-attention block
        self.klens = self.pblock[10 + 2 * B:10 + 3 * B] if not slots else z(B, dt=torch.int32)
        k = int(cfg.ar_kernel)
        self.rings = [z((k - 1) * int(d) + 1, B, D, dt=sdt) for d in cfg.ar_dilations]
        self.hist = z(B, self.max_steps, dt=torch.int32)
        self.ctr = z(8, dt=torch.int32)  # step, -, n_stopped
        self.row_step = z(B, dt=torch.int32)  # per-row copy of the frame index (the sampler's own time base: no ticket)
        self.first_eos = z(B, dt=torch.int32)
        self.stop_t = z(B, dt=torch.int32)
        self.recent = z(B, 64, dt=torch.int32)
        self.owner = None  # weakref to the _ARRun using this plan
        st = hip.ArState()
        st.x_cur = self.x[0].data_ptr()
        st.cond = self.cond.data_ptr()
        st.emb = m.w["cb_embed"].data_ptr()
        st.hist = self.hist.data_ptr()
        st.step = self.ctr.data_ptr()
        st.row_step = self.row_step.data_ptr()
        st.key = self.key.data_ptr()
        st.n_stopped = self.ctr.data_ptr() + 8
        st.first_eos = self.first_eos.data_ptr()
        st.stop_t = self.stop_t.data_ptr()
        st.params = self.params.data_ptr()
        st.recent = self.recent.data_ptr()
        st.nonce = self.nonce.data_ptr()
        st.row_id = self.row_id.data_ptr()
        st.seed = m.seed
        st.B, st.D, st.Tar, st.max_steps, st.V, st.bos_row = B, D, Tar, self.max_steps, m.V, int(cfg.bos_row)
        if slots:  # continuous batching: rows are admitted / released one by one (sopro_amd/continuous.py)
            self.start = torch.full((B,), -1, dtype=torch.int32, device=dev)
            self.row_max = z(B, dt=torch.int32)
            self.row_params = z(B, 8)
            st.start, st.row_max, st.row_params = self.start.data_ptr(), self.row_max.data_ptr(), self.row_params.data_ptr()
        self.state = st
        self.step_t = self.ctr[0:1]
        self.graph: Optional[hip.Graph] = None
        self.nlaunch = 0

    def frame(self) -> "hip.ArFrame":
        """This plan's buffers as the descriptor of sopro_ar_issue_frame (csrc/ar_frame.hip), where the launch sequence of a
        frame lives (reference step: src/sopro/nn/generator.py:98-130)."""
        m, cfg, w, D = self.m, self.m.cfg, self.m.w, self.m.D
        f = hip.ArFrame()
        bf16 = m.precision == "bf16"
        wp = lambda key: hip.ptr(m.wk[key].data, torch.int32 if bf16 else torch.float32)  # noqa: E731
        for i, dil in enumerate(cfg.ar_dilations):
            p, b = f"ar.blocks.{i}", f.blk[i]
            b.glu_w, b.glu_b, b.dw_w, b.dw_b = wp(p + ".glu.w"), hip.ptr(w[p + ".glu.b"]), hip.ptr(w[p + ".dw.w"]), hip.ptr(w[p + ".dw.b"])
            b.ff1_w, b.ff1_b, b.ff2_w, b.ff2_b = wp(p + ".ff1.w"), hip.ptr(w[p + ".ff1.b"]), wp(p + ".ff2.w"), hip.ptr(w[p + ".ff2.b"])
            sdt = torch.bfloat16 if self.bf16_state else torch.float32
            b.ring, b.dil = hip.ptr(self.rings[i], sdt), int(dil)
            if i in self.kp:
                b.xattn, b.gate, b.kp, b.vp = 1, float(m.gates[i]), hip.ptr(self.kp[i], sdt), hip.ptr(self.vp[i], sdt)
                if self.k_unfold:
                    pa = f"ar.x_attns.{i}"
                    b.qa_w, b.qu_w, b.q_b = wp(pa + ".qa.w"), wp(pa + ".qu.w"), hip.ptr(w[pa + ".q.b"])
        f.head_w, f.head_b = wp("ar.head.w"), hip.ptr(w["ar.head.b"])
        X0, XA, XB, _XC = self.x
        f.x0, f.xa, f.xb = hip.ptr(X0), hip.ptr(XA), hip.ptr(XB)
        f.part, f.u, f.xp, f.logits = hip.ptr(self.part), hip.ptr(self.u), hip.ptr(self.xp), hip.ptr(self.logits)
        f.klens = hip.ptr(self.klens, torch.int32)
        f.n_layers, f.B, f.D, f.S_cap, f.V1, f.H, f.ksize = len(cfg.ar_dilations), self.B, D, self.S_cap, m.V + 1, 4, int(cfg.ar_kernel)
        f.w_layout = 2 if bf16 else 1
        f.store_format = 1 if self.bf16_state else 0
        if self.k_unfold:
            f.k_unfold, f.qa, f.qpart = 1, hip.ptr(self.qa), hip.ptr(self.qpart)
        wide = m.ar_tiles_wide if self.B > 32 else None
        f.tile_glu, f.tile_ff1, f.tile_ff2, f.tile_head = (hip.ar_tile_code(wide or m.ar_tiles[k]) for k in ("glu", "ff1", "ff2", "head"))
        f.eps = RMS_EPS
        f.st = self.state
        return f

    def issue_step(self) -> None:
        """Enqueue one frame on the current stream (this is what the graph records)."""
        hip.ar_issue_frame(self.frame())
        self.nlaunch = 3 * len(self.m.cfg.ar_dilations) + len(self.kp) + 2

    def fold_text(self, layer: int, ts: torch.Tensor, nkv: torch.Tensor, kvd: torch.Tensor, *, B: int, S: int, row0: int = 0) -> None:
        """Folded text operands of cross-attention layer ``layer`` for ``B`` utterances starting at plan row ``row0``
        (sopro_ar_fold_text; src/sopro/nn/text.py:75-83 with q_proj / out_proj folded in); in bf16 mode they are folded in fp32
        into the plan's scratch pair and rounded once into the bf16 operands the frame reads."""
        m, w, D = self.m, self.m.w, self.m.D
        pa = f"ar.x_attns.{layer}"
        blk = 4 * self.S_cap * D  # elements per row block
        if self.k_unfold:
            kblk = self.S_cap * D
            if not self.bf16_state:
                hip.ar_fold_text_uk(ts, w[pa + ".nkv.weight"], w[pa + ".kv.w"], w[pa + ".o.w"], nkv, kvd, self.kp[layer], self.vp[layer],
                                    B=B, S=S, S_cap=self.S_cap, D=D, H=4, eps=RMS_EPS, k_off=row0 * kblk, v_off=row0 * blk)
                return
            k32, v32 = self.fold32
            hip.ar_fold_text_uk(ts, w[pa + ".nkv.weight"], w[pa + ".kv.w"], w[pa + ".o.w"], nkv, kvd, k32, v32,
                                B=B, S=S, S_cap=self.S_cap, D=D, H=4, eps=RMS_EPS)
            hip.cvt_f32_bf16(k32, self.kp[layer], n=B * kblk, dst_off=row0 * kblk)
            hip.cvt_f32_bf16(v32, self.vp[layer], n=B * blk, dst_off=row0 * blk)
            return
        if not self.bf16_state:
            hip.ar_fold_text(ts, w[pa + ".nkv.weight"], w[pa + ".kv.w"], w[pa + ".q.wT"], w[pa + ".o.w"], nkv, kvd, self.kp[layer], self.vp[layer],
                             B=B, S=S, S_cap=self.S_cap, D=D, H=4, eps=RMS_EPS, out_off=row0 * blk)
            return
        k32, v32 = self.fold32
        hip.ar_fold_text(ts, w[pa + ".nkv.weight"], w[pa + ".kv.w"], w[pa + ".q.wT"], w[pa + ".o.w"], nkv, kvd, k32, v32,
                         B=B, S=S, S_cap=self.S_cap, D=D, H=4, eps=RMS_EPS, out_off=0)
        hip.cvt_f32_bf16(k32, self.kp[layer], n=B * blk, dst_off=row0 * blk)
        hip.cvt_f32_bf16(v32, self.vp[layer], n=B * blk, dst_off=row0 * blk)

    def load_row(self, row: int, cond_row: torch.Tensor, txt_row: torch.Tensor) -> None:
        """Slot mode: install one utterance in row ``row`` (launches only, on the current stream): its conditioning rows
        [T_r, D], the folded cross-attention operands of its text [S, D] (src/sopro/nn/text.py:75-83), zeroed ring columns."""
        m, cfg, w, D = self.m, self.m.cfg, self.m.w, self.m.D
        Tr, S = int(cond_row.shape[0]), int(txt_row.shape[0])
        if Tr > self.Tar or S > self.S_cap:
            raise ValueError(f"utterance needs {Tr} frames / {S} text positions, the plan holds {self.Tar} / {self.S_cap}")
        self.cond[row, :Tr].copy_(cond_row)
        self.klens[row:row + 1].fill_(S)
        nkv = m.ws.get("ar.row.nkv", (S, D))
        kvd = m.ws.get("ar.row.kvd", (S, 2 * D))
        ts = txt_row.contiguous()
        for i in cfg.ar_xattn_layers:
            self.fold_text(i, ts, nkv, kvd, B=1, S=S, row0=row)
        for r in self.rings:
            r[:, row].zero_()

    def ensure_graph(self, stream: Optional[torch.cuda.Stream] = None) -> None:
        if self.graph is not None or not self.m.use_graph:
            return
        s = stream if stream is not None else self.m.stream
        with torch.cuda.stream(s):
            hip.capture_begin()
            try:
                self.issue_step()
            finally:
                self.graph = hip.capture_end()
            # several frames per replay: the device-side gap between two graph launches (~8 us in a kernel trace) is an order
            # of magnitude above the gap between two nodes of one graph (0.2-0.4 us)
            self.multi = int(hip.dev_env("SOPRO_AR_GRAPH_FRAMES", "8"))
            self.graph_multi = None
            if self.multi > 1:
                hip.capture_begin()
                try:
                    for _ in range(self.multi):
                        self.issue_step()
                finally:
                    self.graph_multi = hip.capture_end()

    def step(self) -> None:
        if self.m.use_graph:
            self.graph.launch()
        else:
            self.issue_step()

    def steps(self, n: int) -> None:
        """n frames on the current stream; with the recorded frame graphs that is one or two calls into the library."""
        n = int(n)
        if self.m.use_graph:
            if getattr(self, "graph_multi", None) is not None and n >= self.multi and not hip.profiling():  # (per-frame events)
                self.graph_multi.launch_n(n // self.multi)
                n %= self.multi
            if n:
                self.graph.launch_n(n)
        else:
            for _ in range(n):
                self.issue_step()


class _ARRun:
    """One batch of utterances being generated on a plan."""

    TOP_K = 50  # hard-coded by the reference's loop (src/sopro/model.py:289-290), like the repetition penalty 1.1

    def __init__(self, m: SoproTTSModel, cond_ar: torch.Tensor, txt_seq: torch.Tensor, text_lens: Optional[torch.Tensor], *,
                 top_p: float, temperature: float, anti_loop: bool, min_gen_frames: Optional[int], seed: Optional[int] = None,
                 top_k: Optional[int] = None, nonces: Optional[Sequence[int]] = None, row_ids: Optional[Sequence[int]] = None,
                 text_lens_host: Optional[Sequence[int]] = None, plan: Optional["_ARPlan"] = None):
        """``nonces`` / ``row_ids`` (one per row, from a scheduler that coalesces several requests into one batch): every row
        then draws what it would have drawn in its own request (nonce of that request, index within it)."""
        import weakref

        cfg, w, dev, D = m.cfg, m.w, m.device, m.D
        if not bool(getattr(cfg, "use_bos", True)):
            raise RuntimeError("BOS embedding disabled")
        top_k = self.TOP_K if top_k is None else int(top_k)
        if not 1 <= top_k <= 64:
            raise ValueError(f"top_k must be in [1, 64] (the reference uses 50), got {top_k}: the device sampler ranks at most 64 entries")
        B, Tar, _ = cond_ar.shape
        S = int(txt_seq.shape[1])
        S_cap = ((S + 63) // 64) * 64
        self.m = m
        self.done = False
        if plan is not None and (plan.B, plan.S_cap, plan.Tar) != (B, S_cap, Tar):
            raise ValueError("the given plan has another shape")
        self.plan = plan = plan if plan is not None else m._ar_plan(B, S_cap, Tar)
        plan.owner = weakref.ref(self)
        min_gen = int(min_gen_frames if min_gen_frames is not None else cfg.min_gen_frames)
        if nonces is not None and len(nonces) != B:
            raise ValueError("one nonce per row")
        with m.on_stream(prep=True):
            # Everything below is a kernel of the library on the preparation stream: nothing of the runtime's (no `copy_`, `fill_`,
            # `zero_`, no host -> device copy) runs between the recorded launch sequences of a pass (round 5).
            if cond_ar.data_ptr() != plan.cond.data_ptr():  # (a scheduler has the conditioning stage write into plan.cond directly)
                hip.copy_u32(plan.cond, cond_ar.to(dev).float().contiguous())
            # the run's parameters: one page-locked block -> the plan's parameter block, one launch
            hb = m._host_block(("ar.params", id(plan)), int(plan.pblock.numel()))
            arr = hb.array()
            arr[0:8].view(np.float32)[:] = [float(top_p), float(temperature), 1.0 if anti_loop else 0.0, 0.85, 1.2, 1.1, float(top_k), float(min_gen)]
            arr[8:10].view(np.uint32)[:] = [m.seed & 0xFFFFFFFF, (m.seed >> 32) & 0xFFFFFFFF]
            arr[10:10 + B].view(np.uint32)[:] = [int(v) & 0xFFFFFFFF for v in nonces] if nonces is not None else m.next_nonce(seed)
            arr[10 + B:10 + 2 * B] = list(row_ids) if row_ids is not None else list(range(B))
            lens_known = text_lens is None or text_lens_host is not None
            arr[10 + 2 * B:10 + 3 * B] = S if text_lens is None else (list(text_lens_host) if text_lens_host is not None else 0)
            hb.copy_to(plan.pblock)
            if not lens_known:  # a device tensor of lengths without its host copy (public ar_generate_batch): device -> device, after the block
                hip.copy_u32(plan.klens, text_lens.to(dev).to(torch.int32).contiguous())
            # K/V of the text for the three cross-attention layers with the query / output projections folded in, once per
            # utterance (src/sopro/nn/text.py:75-83; the sequence is sopro_ar_fold_text, csrc/stages.hip)
            nkv = m.ws.get("ar.nkv", (B * S, D))
            kvd = m.ws.get("ar.kvd", (B * S, 2 * D))
            ts = txt_seq.to(dev).float().contiguous().view(B * S, D)
            for i in cfg.ar_xattn_layers:
                plan.fold_text(i, ts, nkv, kvd, B=B, S=S)
            for r in plan.rings:
                hip.fill_u32(r, 0)
            hip.fill_u32(plan.hist, 0)
            hip.ar_init(plan.state)
            # (the page-locked block is rewritten by the next run on this plan - which starts after this run's tokens were read)
        plan.ensure_graph()
        self._started = False  # the first advance() orders the generation stream behind this preparation

    def _start(self) -> None:
        # here, not in __init__: a scheduler prepares the run OUTSIDE its generation slot and picks the generation stream with
        # the slot (the engine's own stream or a pipeline-fill stream)
        self.m.stream.wait_stream(self.m.prep_stream)
        self._started = True

    def advance(self, n: int) -> None:
        if not self._started:
            self._start()
        with torch.cuda.stream(self.m.stream):
            self.plan.steps(n)  # one call for the whole chunk of frames

    def n_stopped(self, first_eos: bool = False) -> int:
        with torch.cuda.stream(self.m.stream):
            if first_eos:
                v = int((self.plan.first_eos >= 0).sum().item())
            else:
                v = int(self.plan.ctr[2].item())
        return v

    def poll_async(self, first_eos: bool = False):
        """Enqueue a copy of the stop counter behind the launches issued so far; the returned callable waits for it."""
        plan = self.plan
        with torch.cuda.stream(self.m.stream):
            slot = plan.poll_slot = (getattr(plan, "poll_slot", 0) + 1) & 1
            if not hasattr(plan, "poll_host"):
                plan.poll_host = [hip.HostMirror(1) for _ in range(2)]
                plan.poll_ev = [torch.cuda.Event() for _ in range(2)]
            src = (plan.first_eos >= 0).sum().to(torch.int32).view(1) if first_eos else plan.ctr[2:3]
            plan.poll_host[slot].copy_from(src)
            plan.poll_ev[slot].record(self.m.stream)

        def wait() -> int:
            plan.poll_ev[slot].synchronize()
            return int(plan.poll_host[slot].values()[0])

        return wait

    def tokens_host(self, t0: int, t1: int) -> List[int]:
        with torch.cuda.stream(self.m.stream):
            return self.plan.hist[0, t0:t1].tolist()

    def history(self, steps: int, clone: bool = True) -> Tuple[torch.Tensor, List[int]]:
        """Token history [B, steps] and the rows' first-EOS frames (-1: none).  ``clone=False``: a view of the plan's own buffer
        (valid until the next run on this plan starts - a scheduler's refinement reads it in place)."""
        m, plan = self.m, self.plan
        with torch.cuda.stream(m.stream):
            h = plan.hist[:, :steps].clone() if clone else plan.hist[:, :steps]
            hb = m._host_block(("ar.eos", id(plan)), plan.B)
            hb.copy_from(plan.first_eos)  # a kernel of the library writes the page-locked words
        m.stream.synchronize()
        return h, hb.values()
