"""Checkpoint layouts, synthetic checkpoints and safetensors I/O.

Two flat name->tensor dictionaries are understood:

* the Sopro checkpoint: the reference model's ``state_dict`` names
  (reference: src/sopro/model.py:53-117 and the ``nn`` modules it builds; table in
  SURVEY.md Appendix A), stored as one ``model.safetensors`` whose header metadata
  carries the config JSON under ``"cfg"`` (reference: src/sopro/hub.py:30-52);
* the Mimi codec decode side: HuggingFace ``MimiModel.state_dict()`` names
  (third-party transformers/models/mimi/modeling_mimi.py; table in SURVEY.md Appendix B).

There is no network in the build/CI containers, so tests and the benchmark run on
*synthetic* checkpoints: every tensor is drawn from a numpy PCG64 stream keyed by
``(seed, crc32(name))`` so a checkpoint can be regenerated bit-identically anywhere
(the golden fixtures under tests/golden store only the seed).  Parameters the reference
initialises to zero (cross-attention gates, FiLM / adapter output layers, head-id
embeddings, Mimi ``embed_sum``) are de-zeroed, otherwise whole sub-graphs would be
numerically dead and the parity tests vacuous.
"""
from __future__ import annotations

import json
import math
import zlib
from typing import Dict, Iterable, List, Optional, Tuple

import numpy as np

from .config import MimiDecoderConfig, SoproTTSConfig

Spec = Dict[str, Tuple[Tuple[int, ...], str, float]]  # name -> (shape, kind, param)


# ----------------------------------------------------------------------------
# layout of the Sopro checkpoint
# ----------------------------------------------------------------------------
def _ssm_block(spec: Spec, prefix: str, d: int, k: int) -> None:
    """One SSMLiteBlock (reference: src/sopro/nn/blocks.py:113-134)."""
    spec[f"{prefix}.norm.weight"] = ((d,), "norm_w", 0.0)
    spec[f"{prefix}.glu.pro.weight"] = ((2 * d, d), "lin_w", d)
    spec[f"{prefix}.glu.pro.bias"] = ((2 * d,), "lin_b", d)
    spec[f"{prefix}.dw.dw.weight"] = ((d, 1, k), "lin_w", k)
    spec[f"{prefix}.dw.dw.bias"] = ((d,), "lin_b", k)
    spec[f"{prefix}.ff.0.weight"] = ((d,), "norm_w", 0.0)
    spec[f"{prefix}.ff.1.weight"] = ((4 * d, d), "lin_w", d)
    spec[f"{prefix}.ff.1.bias"] = ((4 * d,), "lin_b", d)
    spec[f"{prefix}.ff.3.weight"] = ((d, 4 * d), "lin_w", 4 * d)
    spec[f"{prefix}.ff.3.bias"] = ((d,), "lin_b", 4 * d)


def _xattn_block(spec: Spec, prefix: str, d: int, gate: float) -> None:
    """Text / reference cross-attention block
    (reference: src/sopro/nn/text.py:47-65, src/sopro/nn/ref.py:16-34)."""
    spec[f"{prefix}.gate"] = ((), "const", gate)
    spec[f"{prefix}.nq.weight"] = ((d,), "norm_w", 0.0)
    spec[f"{prefix}.nkv.weight"] = ((d,), "norm_w", 0.0)
    for p in ("q_proj", "k_proj", "v_proj", "out_proj"):
        spec[f"{prefix}.{p}.weight"] = ((d, d), "lin_w", d)


def sopro_weight_spec(cfg: SoproTTSConfig, vocab_size: int) -> Spec:
    """Every persistent tensor of the reference ``SoproTTSModel.state_dict()``."""
    d = int(cfg.d_model)
    q = int(cfg.num_codebooks)
    v = int(cfg.codebook_size)
    spec: Spec = {}

    # text encoder (reference: src/sopro/nn/text.py:16-27)
    spec["text_enc.embed.emb.weight"] = ((int(vocab_size), d), "emb", 1.0)
    for i in range(int(cfg.n_layers_text)):
        _ssm_block(spec, f"text_enc.layers.{i}", d, 7)
    spec["text_enc.norm.weight"] = ((d,), "norm_w", 0.0)

    # shared codebook table, last row = BOS (reference: src/sopro/nn/embeddings.py:37-49)
    spec["cb_embed.emb.weight"] = ((q * v + 1, d), "emb", 1.0)
    spec["nar_prev_cb_weights"] = ((q,), "small", 0.5)

    # Token2SV (reference: src/sopro/nn/speaker.py:12-35)
    sd = 192
    spec["token2sv.cb_weights"] = ((q,), "linspace", 0.0)
    spec["token2sv.emb.weight"] = ((q * v, sd), "emb", 1.0)
    for i in (0, 3):
        spec[f"token2sv.enc.{i}.dw.weight"] = ((sd, 1, 7), "lin_w", 7)
        spec[f"token2sv.enc.{i}.dw.bias"] = ((sd,), "lin_b", 7)
    spec["token2sv.pool.attn.0.weight"] = ((sd, sd), "lin_w", sd)
    spec["token2sv.pool.attn.0.bias"] = ((sd,), "lin_b", sd)
    spec["token2sv.pool.attn.2.weight"] = ((1, sd), "lin_w", sd)
    spec["token2sv.pool.attn.2.bias"] = ((1,), "lin_b", sd)
    svd = int(cfg.sv_student_dim)
    spec["token2sv.proj.weight"] = ((svd, 2 * sd), "lin_w", 2 * sd)
    spec["token2sv.proj.bias"] = ((svd,), "lin_b", 2 * sd)

    # speaker FiLM (reference: src/sopro/nn/speaker.py:64-74); last layer de-zeroed
    spec["spk_film.mlp.0.weight"] = ((d, svd), "lin_w", svd)
    spec["spk_film.mlp.0.bias"] = ((d,), "lin_b", svd)
    spec["spk_film.mlp.2.weight"] = ((2 * d, d), "small", 0.05)
    spec["spk_film.mlp.2.bias"] = ((2 * d,), "small", 0.05)
    spec["spk_film.norm.weight"] = ((d,), "norm_w", 0.0)
    spec["spk_film.norm.bias"] = ((d,), "norm_b", 0.0)

    # AR generator (reference: src/sopro/nn/generator.py:11-42)
    for i in range(int(cfg.n_layers_ar)):
        _ssm_block(spec, f"ar.blocks.{i}", d, int(cfg.ar_kernel))
    for j, i in enumerate(cfg.ar_xattn_layers):
        _xattn_block(spec, f"ar.x_attns.{i}", d, 0.5 + 0.25 * j)
    spec["ar.norm.weight"] = ((d,), "norm_w", 0.0)
    spec["ar.head.weight"] = ((v + 1, d), "lin_w", d / 9.0)
    spec["ar.head.bias"] = ((v + 1,), "lin_b", d)

    # NAR refiner (reference: src/sopro/nn/nar.py:35-87)
    for i in range(int(cfg.n_layers_nar)):
        _ssm_block(spec, f"nar.blocks.{i}", d, int(cfg.nar_kernel_size))
    hd = int(cfg.nar_head_dim)
    spec["nar.norm.weight"] = ((d,), "norm_w", 0.0)
    spec["nar.pre.weight"] = ((hd, d), "lin_w", d)
    spec["nar.pre.bias"] = ((hd,), "lin_b", d)
    stages = cfg.stage_order()
    sc = cfg.stage_codebooks()
    spec["nar.stage_emb.weight"] = ((len(stages), d), "emb", 1.0)
    spec["nar.adapter.norm.weight"] = ((d,), "norm_w", 0.0)
    spec["nar.adapter.mlp.0.weight"] = ((256, d), "lin_w", d)
    spec["nar.adapter.mlp.0.bias"] = ((256,), "lin_b", d)
    spec["nar.adapter.mlp.2.weight"] = ((2 * d, 256), "small", 0.05)
    spec["nar.adapter.mlp.2.bias"] = ((2 * d,), "small", 0.05)
    for s in stages:
        n = len(sc[s])
        for j in range(n):
            spec[f"nar.heads.{s}.{j}.weight"] = ((v, hd), "lin_w", hd / 9.0)
            spec[f"nar.heads.{s}.{j}.bias"] = ((v,), "lin_b", hd)
        spec[f"nar.head_id_emb.{s}.weight"] = ((n, hd), "small", 0.1)
        spec[f"nar.mix.{s}"] = ((2,), "small", 0.5)

    spec["cond_norm.weight"] = ((d,), "norm_w", 0.0)

    # reference encoder + reference cross-attention (reference: src/sopro/model.py:100-117)
    for i in range(int(cfg.ref_enc_layers)):
        _ssm_block(spec, f"ref_enc_blocks.{i}", d, 7)
    spec["ref_enc_norm.weight"] = ((d,), "norm_w", 0.0)
    for i in range(int(cfg.ref_xattn_layers)):
        _xattn_block(spec, f"ref_xattn.blocks.{i}", d, 0.6 + 0.2 * i)
    spec["ref_cb_weights"] = ((q,), "linspace", 0.0)
    return spec


# ----------------------------------------------------------------------------
# layout of the Mimi decode side
# ----------------------------------------------------------------------------
def mimi_decoder_weight_spec(mc: MimiDecoderConfig) -> Spec:
    """Decode-side tensors of HF ``MimiModel.state_dict()``."""
    spec: Spec = {}
    cd, hs = int(mc.codebook_dim), int(mc.hidden_size)
    n_sem = int(mc.num_semantic_quantizers)
    n_ac = int(mc.num_quantizers) - n_sem
    for group, n in (("semantic", n_sem), ("acoustic", n_ac)):
        base = f"quantizer.{group}_residual_vector_quantizer"
        for i in range(n):
            cb = f"{base}.layers.{i}.codebook"
            spec[f"{cb}.embed_sum"] = ((int(mc.codebook_size), cd), "emb", 1.0)
            spec[f"{cb}.cluster_usage"] = ((int(mc.codebook_size),), "usage", 0.0)
            spec[f"{cb}.initialized"] = ((1,), "const", 1.0)
        spec[f"{base}.output_proj.weight"] = ((hs, cd, 1), "lin_w", cd)
    k_up = 2 * int(mc.upsample_stride)
    spec["upsample.conv.weight"] = ((hs, 1, k_up), "lin_w", k_up / 4.0)
    inter = int(mc.intermediate_size)
    for i in range(int(mc.num_hidden_layers)):
        p = f"decoder_transformer.layers.{i}"
        for n in ("q_proj", "k_proj", "v_proj", "o_proj"):
            spec[f"{p}.self_attn.{n}.weight"] = ((hs, hs), "lin_w", hs)
        spec[f"{p}.mlp.fc1.weight"] = ((inter, hs), "lin_w", hs)
        spec[f"{p}.mlp.fc2.weight"] = ((hs, inter), "lin_w", inter)
        for n in ("input_layernorm", "post_attention_layernorm"):
            spec[f"{p}.{n}.weight"] = ((hs,), "norm_w", 0.0)
            spec[f"{p}.{n}.bias"] = ((hs,), "norm_b", 0.0)
        spec[f"{p}.self_attn_layer_scale.scale"] = ((hs,), "scale", 0.25)
        spec[f"{p}.mlp_layer_scale.scale"] = ((hs,), "scale", 0.25)
    # SEANet decoder (modeling_mimi.py:931-961): layer indices count the ELU modules too
    nf = int(mc.num_filters)
    ch = nf * (2 ** len(mc.upsampling_ratios))
    spec["decoder.layers.0.conv.weight"] = ((ch, hs, int(mc.kernel_size)), "lin_w", hs * mc.kernel_size / 3.0)
    spec["decoder.layers.0.conv.bias"] = ((ch,), "lin_b", hs * mc.kernel_size)
    li = 1
    for r in mc.upsampling_ratios:
        li += 1  # ELU
        k = 2 * int(r)
        # ConvTranspose1d weight is (C_in, C_out, k)
        spec[f"decoder.layers.{li}.conv.weight"] = ((ch, ch // 2, k), "lin_w", ch * 2.0 / 3.0)
        spec[f"decoder.layers.{li}.conv.bias"] = ((ch // 2,), "lin_b", ch * 2)
        li += 1
        c = ch // 2
        hdim = c // int(mc.compress)
        rk = int(mc.residual_kernel_size)
        spec[f"decoder.layers.{li}.block.1.conv.weight"] = ((hdim, c, rk), "lin_w", c * rk / 3.0)
        spec[f"decoder.layers.{li}.block.1.conv.bias"] = ((hdim,), "lin_b", c * rk)
        spec[f"decoder.layers.{li}.block.3.conv.weight"] = ((c, hdim, 1), "lin_w", hdim / 3.0)
        spec[f"decoder.layers.{li}.block.3.conv.bias"] = ((c,), "lin_b", hdim)
        li += 1
        ch = c
    li += 1  # final ELU
    lk = int(mc.last_kernel_size)
    spec[f"decoder.layers.{li}.conv.weight"] = ((1, nf, lk), "lin_w", nf * lk / 3.0)
    spec[f"decoder.layers.{li}.conv.bias"] = ((1,), "lin_b", nf * lk)
    return spec


def mimi_encoder_weight_spec(mc: MimiDecoderConfig) -> Spec:
    """Encode-side tensors of HF ``MimiModel.state_dict()``: SEANet encoder (modeling_mimi.py MimiEncoder), encoder
    transformer, downsample conv and the RVQ input projections (the codebooks are shared with the decode side)."""
    spec: Spec = {}
    nf, hs, cd = int(mc.num_filters), int(mc.hidden_size), int(mc.codebook_dim)
    k0, rk, lk = int(mc.kernel_size), int(mc.residual_kernel_size), int(mc.last_kernel_size)
    spec["encoder.layers.0.conv.weight"] = ((nf, 1, k0), "lin_w", k0 / 3.0)
    spec["encoder.layers.0.conv.bias"] = ((nf,), "lin_b", k0)
    li, ch = 1, nf
    for r in reversed(mc.upsampling_ratios):
        hd = ch // int(mc.compress)
        p = f"encoder.layers.{li}.block"
        spec[p + ".1.conv.weight"] = ((hd, ch, rk), "lin_w", ch * rk / 3.0)
        spec[p + ".1.conv.bias"] = ((hd,), "lin_b", ch * rk)
        spec[p + ".3.conv.weight"] = ((ch, hd, 1), "lin_w", hd / 3.0)
        spec[p + ".3.conv.bias"] = ((ch,), "lin_b", hd)
        li += 2  # residual block, ELU
        k = 2 * int(r)
        spec[f"encoder.layers.{li}.conv.weight"] = ((2 * ch, ch, k), "lin_w", ch * k / 3.0)
        spec[f"encoder.layers.{li}.conv.bias"] = ((2 * ch,), "lin_b", ch * k)
        li += 1
        ch *= 2
    li += 1  # ELU
    spec[f"encoder.layers.{li}.conv.weight"] = ((hs, ch, lk), "lin_w", ch * lk / 3.0)
    spec[f"encoder.layers.{li}.conv.bias"] = ((hs,), "lin_b", ch * lk)
    inter = int(mc.intermediate_size)
    for i in range(int(mc.num_hidden_layers)):
        p = f"encoder_transformer.layers.{i}"
        for n in ("q_proj", "k_proj", "v_proj", "o_proj"):
            spec[f"{p}.self_attn.{n}.weight"] = ((hs, hs), "lin_w", hs)
        spec[f"{p}.mlp.fc1.weight"] = ((inter, hs), "lin_w", hs)
        spec[f"{p}.mlp.fc2.weight"] = ((hs, inter), "lin_w", inter)
        for n in ("input_layernorm", "post_attention_layernorm"):
            spec[f"{p}.{n}.weight"] = ((hs,), "norm_w", 0.0)
            spec[f"{p}.{n}.bias"] = ((hs,), "norm_b", 0.0)
        spec[f"{p}.self_attn_layer_scale.scale"] = ((hs,), "scale", 0.25)
        spec[f"{p}.mlp_layer_scale.scale"] = ((hs,), "scale", 0.25)
    spec["downsample.conv.weight"] = ((hs, hs, 2 * int(mc.upsample_stride)), "lin_w", hs * 2 * int(mc.upsample_stride) / 3.0)
    for group in ("semantic", "acoustic"):
        spec[f"quantizer.{group}_residual_vector_quantizer.input_proj.weight"] = ((cd, hs, 1), "lin_w", hs / 3.0)
    return spec


# ----------------------------------------------------------------------------
# synthetic tensors
# ----------------------------------------------------------------------------
def _draw(name: str, shape: Tuple[int, ...], kind: str, param: float, seed: int) -> np.ndarray:
    rng = np.random.Generator(np.random.PCG64([int(seed), zlib.crc32(name.encode())]))
    n = int(np.prod(shape)) if len(shape) else 1
    if kind == "const":
        a = np.full(n, param, dtype=np.float32)
    elif kind == "linspace":
        a = np.linspace(1.0, 0.1, n, dtype=np.float32)
    elif kind in ("lin_w", "lin_b"):
        bound = 1.0 / math.sqrt(max(float(param), 1.0))
        a = rng.uniform(-bound, bound, n).astype(np.float32)
    elif kind == "norm_w":
        a = (1.0 + 0.1 * rng.standard_normal(n)).astype(np.float32)
    elif kind == "norm_b":
        a = (0.1 * rng.standard_normal(n)).astype(np.float32)
    elif kind == "emb":
        a = (param * rng.standard_normal(n)).astype(np.float32)
    elif kind == "small":
        a = (param * rng.standard_normal(n)).astype(np.float32)
    elif kind == "scale":
        a = rng.uniform(0.5 * param, 1.5 * param, n).astype(np.float32)
    elif kind == "usage":
        a = rng.uniform(0.5, 2.0, n).astype(np.float32)
    else:  # pragma: no cover
        raise ValueError(kind)
    return a.reshape(shape)


def synth_tensors(spec: Spec, seed: int) -> Dict[str, np.ndarray]:
    return {name: _draw(name, shp, kind, par, seed) for name, (shp, kind, par) in spec.items()}


def synth_sopro_weights(
    cfg: SoproTTSConfig, vocab_size: int, seed: int = 0, *, suppress_eos: bool = False
) -> Dict[str, np.ndarray]:
    """Seeded synthetic Sopro checkpoint.  ``suppress_eos`` biases the EOS logit to
    -1e9 so that generation always runs to ``max_frames`` (fixed-length benchmark
    runs)."""
    w = synth_tensors(sopro_weight_spec(cfg, vocab_size), seed)
    if suppress_eos:
        w["ar.head.bias"] = w["ar.head.bias"].copy()
        w["ar.head.bias"][cfg.eos_id] = -1e9
    return w


def synth_mimi_weights(mc: MimiDecoderConfig, seed: int = 0, *, with_encoder: bool = False) -> Dict[str, np.ndarray]:
    """Decode-side tensors; ``with_encoder`` adds the encode side (reference audio -> tokens)."""
    w = synth_tensors(mimi_decoder_weight_spec(mc), seed)
    if with_encoder:
        w.update(synth_tensors(mimi_encoder_weight_spec(mc), seed))
    return w


# ----------------------------------------------------------------------------
# safetensors I/O (reference: src/sopro/hub.py:30-52)
# ----------------------------------------------------------------------------
def save_sopro_checkpoint(path: str, weights: Dict[str, np.ndarray], cfg: SoproTTSConfig) -> None:
    from safetensors.numpy import save_file

    # (np.ascontiguousarray would turn the 0-d gate scalars into shape (1,))
    save_file({k: np.require(v, requirements="C") for k, v in weights.items()}, path, metadata={"cfg": cfg.to_json()})


def load_cfg_from_safetensors(path: str) -> SoproTTSConfig:
    from safetensors import safe_open

    with safe_open(path, framework="pt") as f:
        meta = f.metadata() or {}
    if "cfg" not in meta:
        raise RuntimeError(f"No 'cfg' metadata found in {path}.")
    return SoproTTSConfig.from_dict(json.loads(meta["cfg"]))


def load_safetensors(path: str, names: Optional[Iterable[str]] = None) -> Dict[str, np.ndarray]:
    """name -> numpy array; floating tensors of any width (fp16, bf16 - which numpy cannot represent -, fp64) come back as
    float32, integer tensors unchanged.  reference: src/sopro/hub.py:30-52 (safetensors ``load_file`` + ``load_state_dict``)."""
    import torch
    from safetensors import safe_open

    out: Dict[str, np.ndarray] = {}
    with safe_open(path, framework="pt", device="cpu") as f:
        have = set(f.keys())
        keys = sorted(have) if names is None else [k for k in names if k in have]
        for k in keys:
            t = f.get_tensor(k)
            if t.is_floating_point() and t.dtype != torch.float32:
                t = t.to(torch.float32)
            out[k] = t.contiguous().numpy()
    return out


def check_against_spec(weights: Dict[str, np.ndarray], spec: Spec, *, what: str) -> List[str]:
    """Names missing from ``weights`` (shape mismatches raise)."""
    missing: List[str] = []
    for name, (shape, _k, _p) in spec.items():
        if name not in weights:
            missing.append(name)
            continue
        if tuple(weights[name].shape) != tuple(shape):
            raise ValueError(f"{what}: tensor {name} has shape {tuple(weights[name].shape)}, expected {tuple(shape)}")
    return missing


# ----------------------------------------------------------------------------
# badly scaled synthetic checkpoints (range robustness of the split-precision kernels)
# ----------------------------------------------------------------------------
def badly_scaled_sopro(weights: Dict[str, np.ndarray], cfg: SoproTTSConfig, overflow: bool = False) -> Dict[str, np.ndarray]:
    """The same checkpoint with a badly scaled refinement stream.  The reference's arithmetic is range-free fp32
    (src/sopro/nn/blocks.py:26-37, nn/nar.py:89-116); the f16 three-pass kernels have to earn that:
      * the adapter in front of every stage shrinks the stream to ~1e-3 RMS (its ``1 + tanh g`` factor, nar.py:25-32), so the first
        block's fused-RMSNorm contraction stages rows a thousand times smaller than any fixture before;
      * RMSNorm weights x 64 on some blocks and / 64 on others, one block's FF2 x 3000: the stream then climbs to ~1e+3 RMS, and the
        rows the later blocks' contractions stage span six decades over a stage;
      * the rows of ``nar.pre`` are spread over 1e-3 ... 1e+3: the head contraction's operand has that range WITHIN a row.
    ``overflow``: additionally one feed-forward norm weight x 2e5 in all - its GELU output (which the FF2 contraction stages with the
    CONSTANT scale) leaves fp16's range: the kernel's range guard must fire and the engine must fall back to the six-pass operands."""
    w = {k: v.copy() for k, v in weights.items()}
    n, d = int(cfg.n_layers_nar), int(cfg.d_model)
    for i in range(n):
        w[f"nar.blocks.{i}.norm.weight"] *= np.float32((64.0, 1.0 / 64.0, 1.0)[i % 3])
        w[f"nar.blocks.{i}.ff.0.weight"] *= np.float32((1.0 / 64.0, 64.0, 1.0)[i % 3])
    w[f"nar.blocks.{min(2, n - 1)}.ff.3.weight"] *= np.float32(3000.0)
    w["nar.adapter.mlp.2.weight"] *= np.float32(0.003)
    w["nar.adapter.mlp.2.bias"][:d] = np.float32(-3.8)  # 1 + tanh(-3.8) = 1.0e-3
    w["nar.adapter.mlp.2.bias"][d:] *= np.float32(1e-3)
    hd = w["nar.pre.weight"].shape[0]
    rng = np.random.Generator(np.random.PCG64(20260925))
    row_scale = np.logspace(-3.0, 3.0, hd).astype(np.float32)[rng.permutation(hd)]
    w["nar.pre.weight"] *= row_scale[:, None]
    w["nar.pre.bias"] *= row_scale
    if overflow:  # (block 4's norm weight already carries the x 64 of the pattern above: x 2e5 in all, GELU outputs of ~1e5)
        w[f"nar.blocks.{min(4, n - 1)}.ff.0.weight"] *= np.float32(3e3)
    return w


def badly_scaled_mimi(weights: Dict[str, np.ndarray], mc: MimiDecoderConfig) -> Dict[str, np.ndarray]:
    """Decoder-side twin: LayerNorm weights and layer scales of the codec transformer x 16 / / 16 on alternating layers, the SEANet
    transposed convolutions x 4 / / 4 in turn (ELU is not homogeneous: the levels then run in its linear AND its saturated regime)."""
    w = {k: v.copy() for k, v in weights.items()}
    for i in range(int(mc.num_hidden_layers)):
        p = f"decoder_transformer.layers.{i}"
        f = np.float32(16.0 if i % 2 == 0 else 1.0 / 16.0)
        w[f"{p}.input_layernorm.weight"] *= f
        w[f"{p}.self_attn_layer_scale.scale"] /= f
        w[f"{p}.post_attention_layernorm.weight"] /= f
        w[f"{p}.mlp_layer_scale.scale"] *= f
    li = 1
    for si, _r in enumerate(mc.upsampling_ratios):
        li += 1
        w[f"decoder.layers.{li}.conv.weight"] *= np.float32(4.0 if si % 2 == 0 else 0.25)
        li += 2
    return w
