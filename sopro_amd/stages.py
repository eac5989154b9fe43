"""ctypes side of the stage-level C entry points (include/sopro_hip.h: sopro_engine_*, sopro_ar_*, sopro_nar_refine,
sopro_mimi_decode[_stream]).  The launch sequences of the hot path live in the library (csrc/stages.hip, csrc/ar_frame.hip);
``model_engine`` / ``codec_engine`` give the Python host (model.py / codec.py) its engines, and ``StageEngine`` drives all three
stages directly - what a host that is NOT this package writes in ~100 lines (tests/test_gpu_stages.py; INTEGRATION.md shows
the same calls from C)."""
from __future__ import annotations

import ctypes as C
from typing import Dict, Optional, Tuple

import torch

from . import hip

_POS = "BCDEFGHI"


def engine_cfg(cfg, mc, gates: Dict[int, float], nar_mix, prev_w, final_bias: float, rope_positions: int) -> hip.EngineCfg:
    c = hip.EngineCfg()
    c.d_model, c.codebook_size, c.num_codebooks, c.nar_head_dim, c.bos_row = int(cfg.d_model), int(cfg.codebook_size), int(cfg.num_codebooks), \
        int(cfg.nar_head_dim), int(cfg.bos_row)
    c.n_layers_ar, c.ar_kernel = int(cfg.n_layers_ar), int(cfg.ar_kernel)
    for i, d in enumerate(cfg.ar_dilations):
        c.ar_dilations[i] = int(d)
    for i in cfg.ar_xattn_layers:
        c.ar_xattn[i], c.ar_gate[i] = 1, float(gates[i])
    c.n_layers_nar, c.nar_kernel = int(cfg.n_layers_nar), int(cfg.nar_kernel_size)
    for i, d in enumerate(cfg.nar_dilations):
        c.nar_dilations[i] = int(d)
    sc = cfg.stage_codebooks()
    order = cfg.stage_order()
    c.n_stages = len(order)
    for i, s in enumerate(order):
        c.stage_first_cb[i], c.stage_n_cb[i] = int(sc[s][0]), len(sc[s])
        c.nar_mix[i][0], c.nar_mix[i][1] = float(nar_mix[i][0]), float(nar_mix[i][1])
    for i, v in enumerate(prev_w):
        c.nar_prev_cb_weights[i] = float(v)
    c.mimi_hidden, c.mimi_codebook_dim, c.mimi_heads, c.mimi_head_dim = int(mc.hidden_size), int(mc.codebook_dim), int(mc.num_attention_heads), int(mc.head_dim)
    c.mimi_layers, c.mimi_window, c.mimi_inter = int(mc.num_hidden_layers), int(mc.sliding_window), int(mc.intermediate_size)
    c.mimi_n_ratios = len(mc.upsampling_ratios)
    for i, r in enumerate(mc.upsampling_ratios):
        c.mimi_ratios[i] = int(r)
    c.mimi_num_filters, c.mimi_kernel, c.mimi_res_kernel, c.mimi_last_kernel = int(mc.num_filters), int(mc.kernel_size), int(mc.residual_kernel_size), int(mc.last_kernel_size)
    c.mimi_compress, c.mimi_n_semantic, c.mimi_rope_positions = int(mc.compress), int(mc.num_semantic_quantizers), int(rope_positions)
    c.mimi_norm_eps, c.mimi_final_bias = float(mc.norm_eps), float(final_bias)
    return c


class EngineHandle:
    """A ``sopro_engine`` over device tensors that stay owned by the caller (no copies: the engine keeps pointers and builds
    its packed operand forms in ``sopro_engine_finalize``).  ``tensors``: name -> device tensor, the keys of
    ``pack_sopro`` / ``pack_mimi`` (+ "rope.cos" / "rope.sin"); a stage family takes part when its tensors are there."""

    def __init__(self, cfg: hip.EngineCfg, tensors: Dict[str, torch.Tensor], device: torch.device):
        self.lib = lib = hip.load()
        self.device = device
        h = C.c_void_p()
        hip._check(lib.sopro_engine_create(C.byref(cfg), C.byref(h)), "sopro_engine_create")
        self.h = h
        self._keep = []
        for name, t in tensors.items():
            if not (t.is_cuda and t.dtype in (torch.float32, torch.int32) and 1 <= t.dim() <= 4):
                continue
            t = t.contiguous()
            self._keep.append(t)
            shape = (C.c_int64 * t.dim())(*[int(x) for x in t.shape])
            hip._check(lib.sopro_engine_set_tensor(h, name.encode(), t.data_ptr(), shape, t.dim()), "sopro_engine_set_tensor")
        with torch.cuda.device(device):
            st = torch.cuda.Stream(device=device)
            hip._check(lib.sopro_engine_finalize(h, st.cuda_stream), "sopro_engine_finalize")
            st.synchronize()

    def close(self) -> None:
        if self.h:
            torch.cuda.synchronize(self.device)
            self.lib.sopro_engine_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def _sopro_tensors(m) -> Dict[str, torch.Tensor]:
    tensors = dict(m.w)
    tensors["pe"] = m.pe  # position table of the conditioning family
    for i, s in enumerate(m.cfg.stage_order()):  # the C side names the stages by position
        tensors[f"nar.heads.{_POS[i]}.w"], tensors[f"nar.heads.{_POS[i]}.b"] = m.w[f"nar.heads.{s}.w"], m.w[f"nar.heads.{s}.b"]
    return tensors


def _cfg_of(m, codec, precision: str) -> hip.EngineCfg:
    """sopro_engine_cfg from the hosts' configs; a side that is absent contributes its config defaults (unused)."""
    from .config import MimiDecoderConfig, SoproTTSConfig

    scfg = m.cfg if m is not None else SoproTTSConfig()
    mc = codec.mc if codec is not None else MimiDecoderConfig()
    gates = m.gates if m is not None else {i: 0.0 for i in scfg.ar_xattn_layers}
    mix = list(m._nar_mix) if m is not None else [(1.0, 0.0)] * len(scfg.stage_order())
    prev = m.w["nar_prev_cb_weights"].float().cpu().tolist() if m is not None else [0.0] * int(scfg.num_codebooks)
    rope_n = int(codec._rope_tables(1)[0].shape[0]) if codec is not None else 1024
    c = engine_cfg(scfg, mc, gates, mix, prev, codec.final_bias if codec is not None else 0.0, rope_n)
    c.precision = 1 if precision == "bf16" else 0
    c.n_layers_text, c.ref_enc_layers, c.ref_xattn_layers = int(scfg.n_layers_text), int(scfg.ref_enc_layers), int(scfg.ref_xattn_layers)
    c.ref_xattn_heads, c.sv_student_dim, c.enc_kernel = int(scfg.ref_xattn_heads), int(scfg.sv_student_dim), 7
    return c


def model_engine(m) -> EngineHandle:
    """The engine of a ``SoproTTSModel``: AR + NAR families (the lanes of a pipeline share it; it is read-only after finalize)."""
    return EngineHandle(_cfg_of(m, None, m.precision), _sopro_tensors(m), m.device)


def codec_engine(codec) -> EngineHandle:
    """The engine of a ``MimiCodec``: the decoder family + its RoPE tables."""
    tensors = dict(codec.w)
    tensors["rope.cos"], tensors["rope.sin"] = codec._rope_tables(1)
    return EngineHandle(_cfg_of(None, codec, codec.precision), tensors, codec.device)


class StageEngine(EngineHandle):
    """All three stage families in ONE engine over the device weights of an existing ``SoproTTS``, driven directly: what a host
    that is not this package writes (tests/test_gpu_stages.py)."""

    def __init__(self, tts):
        m, codec = tts.model, tts.codec
        self.tts = tts
        tensors = _sopro_tensors(m)
        tensors.update(codec.w)
        tensors["rope.cos"], tensors["rope.sin"] = codec._rope_tables(1)
        super().__init__(_cfg_of(m, codec, m.precision), tensors, m.device)
        self.stream = torch.cuda.Stream(device=self.device)
        self._ws: Dict[str, torch.Tensor] = {}

    def _workspace(self, key: str, nbytes: int) -> torch.Tensor:
        ws = self._ws.get(key)
        if ws is None or ws.numel() < nbytes:
            ws = torch.empty(nbytes, dtype=torch.uint8, device=self.device)
            self._ws[key] = ws
        return ws

    @torch.inference_mode()
    def reference(self, ref_tokens_tq: torch.Tensor) -> Dict[str, object]:
        """One voice from its codec tokens [T, Q] (sopro_ref_prepare): {"sv" [1, svd], "ref_seq" [T, D], "kv" [layers x [T, 2D]]}."""
        cfg, D = self.tts.cfg, int(self.tts.cfg.d_model)
        T, lib, st = int(ref_tokens_tq.shape[0]), self.lib, self.stream
        st.wait_stream(torch.cuda.current_stream(self.device))
        tok = ref_tokens_tq.to(self.device).to(torch.int32).contiguous()
        sv, ref_seq = torch.empty(1, int(cfg.sv_student_dim), device=self.device), torch.empty(T, D, device=self.device)
        kv = [torch.empty(T, 2 * D, device=self.device) for _ in range(int(cfg.ref_xattn_layers))]
        ws = self._workspace("ref", int(lib.sopro_ref_workspace_bytes(self.h, T)))
        with torch.cuda.stream(st):
            hip.ref_prepare(self.h, ws.view(torch.float32), tok, T, sv, ref_seq, kv)
        st.synchronize()
        return {"sv": sv, "ref_seq": ref_seq, "kv": kv}

    @torch.inference_mode()
    def conditioning(self, ids: torch.Tensor, voice: Dict[str, object], max_frames: int, style_strength: float = 1.0) -> Dict[str, torch.Tensor]:
        """One utterance: text ids [S] + a voice of ``reference`` -> {"cond_ar" [1, Tar, D], "txt_seq" [1, S, D]} (sopro_film_coeffs +
        sopro_cond_prepare)."""
        D, lib, st = int(self.tts.cfg.d_model), self.lib, self.stream
        S, Tar = int(ids.numel()), int(max_frames) + 1
        st.wait_stream(torch.cuda.current_stream(self.device))
        idd = ids.to(self.device).to(torch.int32).reshape(1, S).contiguous()
        lens = torch.tensor([S], dtype=torch.int32, device=self.device)
        kvs = voice["kv"]
        Tr = int(kvs[0].shape[0])
        ks, vs = [kv[:, :D].contiguous() for kv in kvs], [kv[:, D:].contiguous() for kv in kvs]
        mul, add = torch.empty(1, D, device=self.device), torch.empty(1, D, device=self.device)
        txt_seq, txt_pool = torch.empty(1, S, D, device=self.device), torch.empty(1, D, device=self.device)
        cond_ar = torch.empty(1, Tar, D, device=self.device)
        ws = self._workspace("cond", int(lib.sopro_cond_workspace_bytes(self.h, 1, S, Tar)))
        with torch.cuda.stream(st):
            hip.film_coeffs(self.h, voice["sv"], float(style_strength), 1, torch.empty(5 * D, device=self.device), mul, add)
            hip.cond_prepare(self.h, ws.view(torch.float32), idd, lens, False, mul, add, ks, vs, Tr * D, None, None, 1, S, Tar, Tr, txt_seq, txt_pool, cond_ar)
        st.synchronize()
        return {"cond_ar": cond_ar, "txt_seq": txt_seq}

    @torch.inference_mode()
    def ar_generate(self, cond_ar: torch.Tensor, txt_seq: torch.Tensor, text_lens: Optional[torch.Tensor], *, top_p: float, temperature: float,
                    anti_loop: bool, min_gen_frames: int = 12, seed: int = 0, nonce: int = 0, steps: Optional[int] = None
                    ) -> Tuple[torch.Tensor, torch.Tensor]:
        """-> (hist [B, Tar] int32, first_eos [B] int32) on the device."""
        B, Tar, D = cond_ar.shape
        S = int(txt_seq.shape[1])
        lib, st = self.lib, self.stream
        st.wait_stream(torch.cuda.current_stream(self.device))
        ws = self._workspace("ar", int(lib.sopro_ar_workspace_bytes(self.h, B, S, Tar)))
        cond_ar, txt_seq = cond_ar.float().contiguous(), txt_seq.float().contiguous()
        lens = text_lens.to(torch.int32).contiguous() if text_lens is not None else None
        prm = (C.c_float * 8)(float(top_p), float(temperature), 1.0 if anti_loop else 0.0, 0.85, 1.2, 1.1, 50.0, float(min_gen_frames))
        hist = torch.empty(B, Tar, dtype=torch.int32, device=self.device)
        feos = torch.empty(B, dtype=torch.int32, device=self.device)
        hip._check(lib.sopro_ar_begin(self.h, ws.data_ptr(), B, cond_ar.data_ptr(), txt_seq.data_ptr(), lens.data_ptr() if lens is not None else None,
                                      S, Tar, prm, int(seed), int(nonce), st.cuda_stream), "sopro_ar_begin")
        hip._check(lib.sopro_ar_run_graph(self.h, Tar if steps is None else int(steps), st.cuda_stream), "sopro_ar_run_graph")
        hip._check(lib.sopro_ar_tokens(self.h, hist.data_ptr(), feos.data_ptr(), None, st.cuda_stream), "sopro_ar_tokens")
        st.synchronize()
        return hist, feos

    @torch.inference_mode()
    def nar_refine(self, cond: torch.Tensor, rvq1: torch.Tensor, lens: Optional[torch.Tensor] = None) -> torch.Tensor:
        """cond [B, T', D] (T' >= T rows per utterance), rvq1 [B, T] -> tokens [B, T, Q] int32."""
        B, T = rvq1.shape
        lib, st = self.lib, self.stream
        st.wait_stream(torch.cuda.current_stream(self.device))
        ws = self._workspace("nar", int(lib.sopro_nar_workspace_bytes(self.h, B, T)))
        cond = cond.float()
        assert cond.stride(2) == 1 and cond.stride(1) == cond.shape[2]
        rvq1 = rvq1.to(torch.int32).contiguous()
        lens_d = lens.to(torch.int32).contiguous() if lens is not None else None
        out = torch.empty(B, T, int(self.tts.cfg.num_codebooks), dtype=torch.int32, device=self.device)
        hip._check(lib.sopro_nar_refine(self.h, ws.data_ptr(), cond.data_ptr(), int(cond.stride(0)), rvq1.data_ptr(),
                                        lens_d.data_ptr() if lens_d is not None else None, B, T, out.data_ptr(), st.cuda_stream), "sopro_nar_refine")
        st.synchronize()
        return out

    @torch.inference_mode()
    def nar_refine_guarded(self, cond: torch.Tensor, rvq1: torch.Tensor, lens: Optional[torch.Tensor] = None) -> Tuple[torch.Tensor, int]:
        """The refinement with its RANGE GUARD, as a C host writes it (sopro_nar_refine_io): the pass leaves the number of workgroups
        that had to saturate an fp16 operand in a word of the caller's; a non-zero word means the tokens cannot be trusted, and the
        host repeats the pass with ``safe = 1`` (six-pass bf16 operands: fp32's exponent range).  -> (tokens [B, T, Q] int32, events
        of the first pass)."""
        B, T = rvq1.shape
        lib, st = self.lib, self.stream
        st.wait_stream(torch.cuda.current_stream(self.device))
        ws = self._workspace("nar", int(lib.sopro_nar_workspace_bytes(self.h, B, T)))
        cond = cond.float()
        assert cond.stride(2) == 1 and cond.stride(1) == cond.shape[2]
        rvq1 = rvq1.to(torch.int32).contiguous()
        lens_d = (lens if lens is not None else torch.full((B,), T)).to(self.device).to(torch.int32).contiguous()
        out = torch.empty(B, T, int(self.tts.cfg.num_codebooks), dtype=torch.int32, device=self.device)
        word = torch.zeros(1, dtype=torch.int32, device=self.device)
        io = hip.NarIO(cond.data_ptr(), int(cond.stride(0)), rvq1.data_ptr(), int(rvq1.stride(0)), lens_d.data_ptr(), out.data_ptr(), word.data_ptr(), 0)
        with torch.cuda.stream(st):
            hip._check(lib.sopro_nar_refine_io(self.h, ws.data_ptr(), C.byref(io), B, T, st.cuda_stream), "sopro_nar_refine_io")
            st.synchronize()
            events = int(word.item())
            if events:
                io.safe = 1
                word.zero_()
                hip._check(lib.sopro_nar_refine_io(self.h, ws.data_ptr(), C.byref(io), B, T, st.cuda_stream), "sopro_nar_refine_io")
                st.synchronize()
                assert int(word.item()) == 0, "the six-pass operands have fp32's range: no events"
        return out, events

    @torch.inference_mode()
    def mimi_decode(self, tokens: torch.Tensor) -> torch.Tensor:
        """tokens [B, T, Q] -> wav [B, T * 1920]."""
        B, T, _ = tokens.shape
        lib, st = self.lib, self.stream
        st.wait_stream(torch.cuda.current_stream(self.device))
        ws = self._workspace("mimi", int(lib.sopro_mimi_workspace_bytes(self.h, B, T)))
        tok = tokens.to(torch.int32).contiguous()
        wav = torch.empty(B, T * int(self.tts.codec.mc.frame_samples), device=self.device)
        hip._check(lib.sopro_mimi_decode(self.h, ws.data_ptr(), tok.data_ptr(), B, T, wav.data_ptr(), st.cuda_stream), "sopro_mimi_decode")
        st.synchronize()
        return wav


class CheckpointEngine(StageEngine):
    """The same driver over an engine the LIBRARY built from the reference's own files (csrc/checkpoint.hip:
    sopro_checkpoint_open + sopro_engine_from_checkpoint): model.safetensors with SoproTTSModel.state_dict() keys and the Mimi
    codec's safetensors with HuggingFace keys - no pack.py, no weights.py, no torch tensor of a weight on the Python side.  What
    INTEGRATION.md's C host does; tests/test_gpu_stages.py runs the reference's 200-frame fixture on it."""

    def __init__(self, sopro_path: str, mimi_path: str, device: str = "cuda:0", precision: str = "f32"):
        import types

        from .config import MimiDecoderConfig
        from .weights import load_cfg_from_safetensors

        self.lib = lib = hip.load()
        self.device = torch.device(device)
        cfg = load_cfg_from_safetensors(sopro_path)  # (shapes of the driver's own buffers only)
        self.tts = types.SimpleNamespace(cfg=cfg, codec=types.SimpleNamespace(mc=MimiDecoderConfig(num_quantizers=int(cfg.num_codebooks))))
        ck = C.c_void_p()
        hip._check(lib.sopro_checkpoint_open(sopro_path.encode(), mimi_path.encode(), C.byref(ck)), "sopro_checkpoint_open")
        try:
            with torch.cuda.device(self.device):
                self.stream = torch.cuda.Stream(device=self.device)
                h = C.c_void_p()
                hip._check(lib.sopro_engine_from_checkpoint(ck, 1 if precision == "bf16" else 0, self.stream.cuda_stream, C.byref(h)),
                           "sopro_engine_from_checkpoint")
                self.stream.synchronize()
        finally:
            lib.sopro_checkpoint_close(ck)
        self.h = h
        self._keep = []
        self._ws = {}


class StageStreamDecoder:
    """MimiStreamDecoder.decode_step (src/sopro/codec/mimi.py:115-181) over ``sopro_mimi_decode_stream``: the 2-frame token
    overlap and the cropping are the host's few lines, the decode with the cached keys / values is the C call."""

    def __init__(self, eng: StageEngine, overlap_frames: int = 2, trim: str = "none", cap_rows: int = 4096):
        self.eng, self.ov, self.trim = eng, int(overlap_frames), trim
        lib = eng.lib
        self.kv = torch.empty(int(lib.sopro_mimi_stream_kv_bytes(eng.h, cap_rows)), dtype=torch.uint8, device=eng.device)
        self.st = hip.MimiStreamState()
        hip._check(lib.sopro_mimi_stream_init(eng.h, C.byref(self.st), self.kv.data_ptr(), cap_rows), "sopro_mimi_stream_init")
        self.tail: Optional[torch.Tensor] = None

    @torch.inference_mode()
    def decode_step(self, codes_chunk_tq: torch.Tensor) -> torch.Tensor:
        eng, lib = self.eng, self.eng.lib
        hop = int(eng.tts.codec.mc.frame_samples)
        chunk = codes_chunk_tq.to(eng.device).to(torch.int32)
        n_new = int(chunk.shape[0])
        ov = 0
        codes_in = chunk
        if self.ov > 0 and self.tail is not None and self.tail.numel() > 0:
            ov = min(self.ov, int(self.tail.shape[0]))
            codes_in = torch.cat([self.tail[-ov:], chunk], dim=0)
        if self.trim == "legacy" and ov > 0:
            hip._check(lib.sopro_mimi_stream_trim(C.byref(self.st), ov), "sopro_mimi_stream_trim")
        T = int(codes_in.shape[0])
        st = eng.stream
        st.wait_stream(torch.cuda.current_stream(eng.device))
        ws = eng._workspace("mimi", int(lib.sopro_mimi_workspace_bytes(eng.h, 1, T)))
        codes_in = codes_in.contiguous()
        wav = torch.empty(1, T * hop, device=eng.device)
        hip._check(lib.sopro_mimi_decode_stream(eng.h, ws.data_ptr(), C.byref(self.st), codes_in.data_ptr(), T, wav.data_ptr(), st.cuda_stream),
                   "sopro_mimi_decode_stream")
        st.synchronize()
        self.tail = codes_in[-min(self.ov, T):].clone() if self.ov > 0 else None
        return wav[:, : (ov + n_new) * hop][:, ov * hop:]
