"""Model hyper-parameter schema for the Sopro hot path.

The field names and defaults are the reference's checkpoint schema
(reference: src/sopro/config.py:7-43) because real checkpoints carry this dict as
JSON in the safetensors header (reference: src/sopro/hub.py:30-48).  Everything
else in this module (derived geometry, Mimi decoder geometry, engine options)
is new and only describes what the HIP engine needs to size its buffers.
"""
from __future__ import annotations

import dataclasses
import json
from dataclasses import dataclass, field
from typing import Any, Dict, List, Tuple

TARGET_SR = 24000  # reference: src/sopro/constants.py:3
DEFAULT_MIMI_ID = "kyutai/mimi"  # reference: src/sopro/constants.py:5
FRAME_SAMPLES = 1920  # 24 kHz / 12.5 fps


@dataclass
class SoproTTSConfig:
    # -- codec geometry
    num_codebooks: int = 32
    codebook_size: int = 2048
    mimi_fps: float = 12.5
    max_frames: int = 400
    audio_sr: int = TARGET_SR
    # -- trunk
    d_model: int = 384
    n_layers_text: int = 2
    dropout: float = 0.05
    pos_emb_max: int = 4096
    max_text_len: int = 2048
    # -- autoregressive generator (codebook 0)
    n_layers_ar: int = 6
    ar_kernel: int = 13
    ar_dilation_cycle: Tuple[int, ...] = (1, 2, 4, 1)
    ar_text_attn_freq: int = 2
    min_gen_frames: int = 12
    # -- non-autoregressive refiner (codebooks 1..Q-1)
    n_layers_nar: int = 6
    nar_head_dim: int = 256
    nar_kernel_size: int = 11
    nar_dilation_cycle: Tuple[int, ...] = (1, 2, 4, 8)
    stage_B: Tuple[int, int] = (2, 4)
    stage_C: Tuple[int, int] = (5, 8)
    stage_D: Tuple[int, int] = (9, 16)
    stage_E: Tuple[int, int] = (17, 32)
    # -- speaker / reference conditioning
    sv_student_dim: int = 192
    style_strength: float = 1.0
    ref_enc_layers: int = 2
    ref_xattn_heads: int = 2
    ref_xattn_layers: int = 3
    ref_xattn_gmax: float = 0.35

    # ---- construction helpers -------------------------------------------------
    @classmethod
    def from_dict(cls, d: Dict[str, Any]) -> "SoproTTSConfig":
        """Key-intersection load, like the reference's checkpoint reader
        (reference: src/sopro/hub.py:44-48): unknown keys are ignored,
        missing keys keep their defaults, lists become tuples."""
        known = {f.name: f for f in dataclasses.fields(cls)}
        kw: Dict[str, Any] = {}
        for k, v in d.items():
            if k not in known:
                continue
            if isinstance(v, list):
                v = tuple(v)
            kw[k] = v
        return cls(**kw)

    def to_json(self) -> str:
        return json.dumps(dataclasses.asdict(self))

    # ---- derived geometry (what the engine sizes its state from) --------------
    @staticmethod
    def _cycle(cycle: Tuple[int, ...], n: int) -> Tuple[int, ...]:
        cyc = tuple(int(x) for x in cycle) or (1,)
        out: List[int] = []
        while len(out) < n:
            out.extend(cyc)
        return tuple(out[:n])

    @property
    def ar_dilations(self) -> Tuple[int, ...]:
        # reference: src/sopro/nn/generator.py:16-20
        return self._cycle(self.ar_dilation_cycle, int(self.n_layers_ar))

    @property
    def nar_dilations(self) -> Tuple[int, ...]:
        # reference: src/sopro/nn/nar.py:49-54
        return self._cycle(self.nar_dilation_cycle, int(self.n_layers_nar))

    @property
    def ar_xattn_layers(self) -> Tuple[int, ...]:
        # text cross-attention after every ar_text_attn_freq-th block
        # (reference: src/sopro/nn/generator.py:31-39)
        f = int(self.ar_text_attn_freq)
        return tuple(i for i in range(int(self.n_layers_ar)) if (i + 1) % f == 0)

    @property
    def eos_id(self) -> int:
        return int(self.codebook_size)  # reference: src/sopro/model.py:59

    @property
    def bos_row(self) -> int:
        # last row of the shared codebook table (reference: src/sopro/nn/embeddings.py:47-49)
        return int(self.num_codebooks) * int(self.codebook_size)

    def stage_codebooks(self) -> Dict[str, List[int]]:
        """Stage name -> 0-based codebook columns it predicts
        (reference: src/sopro/model.py:39-42,85-94)."""
        q = int(self.num_codebooks)
        out: Dict[str, List[int]] = {}
        for name in ("B", "C", "D", "E"):
            lo, hi = getattr(self, "stage_" + name)
            out[name] = [i for i in range(int(lo) - 1, int(hi)) if 1 <= i < q]
        return out

    def stage_order(self) -> List[str]:
        sc = self.stage_codebooks()
        return [s for s in ("B", "C", "D", "E") if len(sc[s]) > 0]

    def rf_ar(self) -> int:
        # reference: src/sopro/sampling.py:96-97
        return 1 + (int(self.ar_kernel) - 1) * sum(self.ar_dilations)

    def rf_nar(self) -> int:
        # reference: src/sopro/sampling.py:100-101, src/sopro/model.py:125-131
        return 1 + (int(self.nar_kernel_size) - 1) * sum(self.nar_dilations)


@dataclass
class MimiDecoderConfig:
    """Decode-side geometry of the Mimi codec (third-party: HuggingFace
    transformers `MimiConfig` defaults, transformers/models/mimi/configuration_mimi.py:86-133,
    with num_quantizers overridden by the Sopro checkpoint as the reference does at
    src/sopro/codec/mimi.py:28-31)."""

    num_quantizers: int = 32
    num_semantic_quantizers: int = 1
    codebook_size: int = 2048
    codebook_dim: int = 256
    hidden_size: int = 512
    num_filters: int = 64
    kernel_size: int = 7
    last_kernel_size: int = 3
    residual_kernel_size: int = 3
    compress: int = 2
    upsampling_ratios: Tuple[int, ...] = (8, 6, 5, 4)
    num_hidden_layers: int = 8
    num_attention_heads: int = 8
    head_dim: int = 64
    intermediate_size: int = 2048
    norm_eps: float = 1e-5
    sliding_window: int = 250
    rope_theta: float = 10000.0
    upsample_stride: int = 2  # 12.5 Hz -> 25 Hz depthwise ConvTranspose1d, k = 2*stride
    sampling_rate: int = TARGET_SR
    frame_rate: float = 12.5

    @property
    def frame_samples(self) -> int:
        return int(round(self.sampling_rate / self.frame_rate))


@dataclass
class EngineOptions:
    """New knobs of the MI355X engine (not part of the reference API)."""

    weight_dtype: str = "f32"  # "f32" (parity mode, exact-f32 MFMA) or "bf16"
    use_graph: bool = True  # capture the per-frame AR step in a hipGraph
    graph_steps: int = 1  # AR steps per captured graph
    seed: int = 0  # Philox seed of the on-device sampler
    extra: Dict[str, Any] = field(default_factory=dict)
