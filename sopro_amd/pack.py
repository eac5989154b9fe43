"""Host-side weight repacking: reference checkpoint layouts -> the [N, K] row-major fp32 operands
the HIP kernels read (include/sopro_hip.h).  Runs once at load time, on the CPU, in torch;
nothing here is on the timed path.

Conventions (channels-last activations):
  * GLU projections of full-sequence blocks are packed per 64 rows as [32 value | 32 gate] so the
    GEMM epilogue can gate in registers (reference: src/sopro/nn/blocks.py:16-23);
  * depthwise taps become tap-major [k, C] (reference layout [C, 1, k], blocks.py:45-49);
  * Conv1d weights [Cout, Cin, k] become [Cout, k*Cin] (tap-major K), so that a causal conv is a
    contraction over k consecutive channels-last rows (HF:modeling_mimi.py:210-347);
  * ConvTranspose1d weights [Cin, Cout, 2s] (stride s) become [s*Cout, 2*Cin]: output row t holds
    the s samples t*s..t*s+s-1, input row is [x[t-1] | x[t]] (HF:modeling_mimi.py:350-405).
"""
from __future__ import annotations

from typing import Dict

import numpy as np
import torch

from .config import MimiDecoderConfig, SoproTTSConfig


def _t(a) -> torch.Tensor:
    t = a if isinstance(a, torch.Tensor) else torch.from_numpy(np.ascontiguousarray(a))
    return t.float() if t.is_floating_point() else t


def pack_glu(w: torch.Tensor, b: torch.Tensor):
    """[2D, K] (value rows then gate rows) -> per-64 blocks [32 value | 32 gate]."""
    d = w.shape[0] // 2
    assert d % 32 == 0
    wv, wg = w[:d].reshape(d // 32, 32, -1), w[d:].reshape(d // 32, 32, -1)
    wp = torch.cat([wv, wg], dim=1).reshape(2 * d, -1).contiguous()
    bv, bg = b[:d].reshape(d // 32, 32), b[d:].reshape(d // 32, 32)
    bp = torch.cat([bv, bg], dim=1).reshape(2 * d).contiguous()
    return wp, bp


def pack_dw(w: torch.Tensor) -> torch.Tensor:
    """[C, 1, k] -> [k, C]"""
    return w.squeeze(1).t().contiguous()


def pack_conv1d(w: torch.Tensor) -> torch.Tensor:
    """[Cout, Cin, k] -> [Cout, k*Cin] with K index = tap*Cin + ci."""
    co, ci, k = w.shape
    return w.permute(0, 2, 1).reshape(co, k * ci).contiguous()


def pack_convtr1d(w: torch.Tensor, b: torch.Tensor, stride: int):
    """[Cin, Cout, 2s] -> ([s*Cout, 2*Cin], [s*Cout]); row r*Cout+co, col half*Cin+ci = w[ci, co, (1-half)*s + r]."""
    ci, co, k = w.shape
    assert k == 2 * stride
    wv = w.reshape(ci, co, 2, stride)  # [ci, co, h2, r], tap = h2*s + r ; h2 = 0 multiplies x[t], h2 = 1 multiplies x[t-1]
    wp = wv.flip(2).permute(3, 1, 2, 0).reshape(stride * co, 2 * ci).contiguous()
    bp = b.repeat(stride).contiguous()
    return wp, bp


def _ssm_block(out: Dict[str, torch.Tensor], w: Dict[str, torch.Tensor], p: str, *, packed_glu: bool) -> None:
    out[p + ".norm.weight"] = w[p + ".norm.weight"]
    if packed_glu:
        out[p + ".glu.w"], out[p + ".glu.b"] = pack_glu(w[p + ".glu.pro.weight"], w[p + ".glu.pro.bias"])
    else:
        out[p + ".glu.w"], out[p + ".glu.b"] = w[p + ".glu.pro.weight"], w[p + ".glu.pro.bias"]
    out[p + ".dw.w"] = pack_dw(w[p + ".dw.dw.weight"])
    out[p + ".dw.b"] = w[p + ".dw.dw.bias"]
    out[p + ".ff.norm.weight"] = w[p + ".ff.0.weight"]
    out[p + ".ff1.w"], out[p + ".ff1.b"] = w[p + ".ff.1.weight"], w[p + ".ff.1.bias"]
    out[p + ".ff2.w"], out[p + ".ff2.b"] = w[p + ".ff.3.weight"], w[p + ".ff.3.bias"]


def _xattn(out: Dict[str, torch.Tensor], w: Dict[str, torch.Tensor], p: str, gate_mul: float, heads: int = 0) -> None:
    d = w[p + ".q_proj.weight"].shape[0]
    out[p + ".nq.weight"] = w[p + ".nq.weight"]
    out[p + ".nkv.weight"] = w[p + ".nkv.weight"]
    out[p + ".q.w"] = w[p + ".q_proj.weight"]
    out[p + ".kv.w"] = torch.cat([w[p + ".k_proj.weight"], w[p + ".v_proj.weight"]], dim=0).contiguous()
    out[p + ".o.w"] = w[p + ".out_proj.weight"]
    # per-head transposed query projection [H, d, dh]: folds q_proj into the cached keys (K'_h = K_h Wq_h)
    if heads:
        out[p + ".q.wT"] = w[p + ".q_proj.weight"].reshape(heads, d // heads, d).permute(0, 2, 1).contiguous()
    # x + (gmax *) tanh(gate) * a  -> per-column scale of the residual epilogue
    out[p + ".gate_scale"] = (gate_mul * torch.tanh(w[p + ".gate"].float())).reshape(1).repeat(d).contiguous()


def pack_sopro(weights: Dict[str, "np.ndarray"], cfg: SoproTTSConfig) -> Dict[str, torch.Tensor]:
    """Reference ``SoproTTSModel.state_dict()`` names -> kernel operands (CPU tensors)."""
    w = {k: _t(v) for k, v in weights.items()}
    out: Dict[str, torch.Tensor] = {}
    for i in range(int(cfg.n_layers_text)):
        _ssm_block(out, w, f"text_enc.layers.{i}", packed_glu=True)
    out["text_enc.embed"] = w["text_enc.embed.emb.weight"]
    out["text_enc.norm.weight"] = w["text_enc.norm.weight"]
    out["cb_embed"] = w["cb_embed.emb.weight"]
    out["nar_prev_cb_weights"] = w["nar_prev_cb_weights"]
    # Token2SV (reference: src/sopro/nn/speaker.py:12-61)
    out["token2sv.emb"] = w["token2sv.emb.weight"]
    out["token2sv.cw"] = torch.softmax(w["token2sv.cb_weights"].float(), dim=0)
    for i in (0, 3):
        out[f"token2sv.enc.{i}.w"] = pack_dw(w[f"token2sv.enc.{i}.dw.weight"])
        out[f"token2sv.enc.{i}.b"] = w[f"token2sv.enc.{i}.dw.bias"]
    for n in ("pool.attn.0", "pool.attn.2", "proj"):
        out[f"token2sv.{n}.w"] = w[f"token2sv.{n}.weight"]
        out[f"token2sv.{n}.b"] = w[f"token2sv.{n}.bias"]
    # speaker FiLM
    for n in ("mlp.0", "mlp.2"):
        out[f"spk_film.{n}.w"], out[f"spk_film.{n}.b"] = w[f"spk_film.{n}.weight"], w[f"spk_film.{n}.bias"]
    out["spk_film.norm.weight"], out["spk_film.norm.bias"] = w["spk_film.norm.weight"], w["spk_film.norm.bias"]
    # AR generator: natural GLU layout (the step kernel pairs value/gate rows itself)
    # The step kernels apply the RMSNorm row scale to the accumulator, so the norm's weight vector is folded into the
    # following projection here: (x * rstd * w_norm) @ W^T == rstd * (x @ (W * w_norm)^T)   (blocks.py:26-37)
    for i in range(int(cfg.n_layers_ar)):
        p = f"ar.blocks.{i}"
        _ssm_block(out, w, p, packed_glu=False)
        out[p + ".glu.w"] = (out[p + ".glu.w"] * w[p + ".norm.weight"][None, :]).contiguous()
        out[p + ".ff1.w"] = (out[p + ".ff1.w"] * w[p + ".ff.0.weight"][None, :]).contiguous()
    for i in cfg.ar_xattn_layers:
        p = f"ar.x_attns.{i}"
        _xattn(out, w, p, 1.0, heads=4)  # 4 heads: reference src/sopro/nn/generator.py:36
        out[p + ".q.wT"] = (out[p + ".q.wT"] * w[p + ".nq.weight"][None, :, None]).contiguous()  # fold RMSNorm_nq's weight
        # Unfolded keys (sopro_ar_frame.k_unfold): the query q_raw = Wq' x with Wq' = q_proj * RMSNorm_nq's weight is linear in the
        # residual stream x = out + b2 + W2 u of block i, so it is emitted by that block's feed-forward launches:
        #   qa.w = Wq' (on the rows FF1 stages), qu.w = Wq' W2 and q.b = Wq' b2 (on FF2's K-slices); products in float64
        wq = (w[p + ".q_proj.weight"].double() * w[p + ".nq.weight"].double()[None, :])
        out[p + ".qa.w"] = wq.float().contiguous()
        out[p + ".qu.w"] = (wq @ w[f"ar.blocks.{i}.ff.3.weight"].double()).float().contiguous()
        out[p + ".q.b"] = (wq @ w[f"ar.blocks.{i}.ff.3.bias"].double()).float().contiguous()
    out["ar.norm.weight"] = w["ar.norm.weight"]
    out["ar.head.w"] = (w["ar.head.weight"] * w["ar.norm.weight"][None, :]).contiguous()
    out["ar.head.b"] = w["ar.head.bias"]
    # NAR refiner
    for i in range(int(cfg.n_layers_nar)):
        _ssm_block(out, w, f"nar.blocks.{i}", packed_glu=True)
    out["nar.norm.weight"] = w["nar.norm.weight"]
    out["nar.pre.w"], out["nar.pre.b"] = w["nar.pre.weight"], w["nar.pre.bias"]
    out["nar.stage_emb"] = w["nar.stage_emb.weight"]
    out["nar.adapter.norm.weight"] = w["nar.adapter.norm.weight"]
    for n in ("mlp.0", "mlp.2"):
        out[f"nar.adapter.{n}.w"], out[f"nar.adapter.{n}.b"] = w[f"nar.adapter.{n}.weight"], w[f"nar.adapter.{n}.bias"]
    sc = cfg.stage_codebooks()
    for s in cfg.stage_order():
        # The heads of a stage become ONE projection [nh*V, HD]: logits_j = (z + e_j) W_j^T + b_j = z W_j^T + (b_j + W_j e_j)
        # (src/sopro/nn/nar.py:100-116), the head-id embedding e_j folded into the bias in float64.
        hid = w[f"nar.head_id_emb.{s}.weight"].double()
        ws_, bs_ = [], []
        for j in range(len(sc[s])):
            wj, bj = w[f"nar.heads.{s}.{j}.weight"], w[f"nar.heads.{s}.{j}.bias"]
            ws_.append(wj)
            bs_.append((bj.double() + wj.double() @ hid[j]).float())
        out[f"nar.heads.{s}.w"], out[f"nar.heads.{s}.b"] = torch.cat(ws_, dim=0).contiguous(), torch.cat(bs_, dim=0).contiguous()
        out[f"nar.mix.{s}"] = torch.softmax(w[f"nar.mix.{s}"].float(), dim=0)
    out["cond_norm.weight"] = w["cond_norm.weight"]
    # reference encoder + reference cross-attention
    for i in range(int(cfg.ref_enc_layers)):
        _ssm_block(out, w, f"ref_enc_blocks.{i}", packed_glu=True)
    out["ref_enc_norm.weight"] = w["ref_enc_norm.weight"]
    out["ref_cw"] = torch.softmax(w["ref_cb_weights"].float(), dim=0)
    for i in range(int(cfg.ref_xattn_layers)):
        _xattn(out, w, f"ref_xattn.blocks.{i}", float(cfg.ref_xattn_gmax))
    return {k: v.contiguous() for k, v in out.items()}


def pack_mimi(weights: Dict[str, "np.ndarray"], mc: MimiDecoderConfig) -> Dict[str, torch.Tensor]:
    """HF ``MimiModel.state_dict()`` decode-side names -> kernel operands (CPU tensors)."""
    w = {k: _t(v) for k, v in weights.items()}
    out: Dict[str, torch.Tensor] = {}
    ns = int(mc.num_semantic_quantizers)
    tabs = []
    for q in range(int(mc.num_quantizers)):
        grp, i = ("semantic", q) if q < ns else ("acoustic", q - ns)
        p = f"quantizer.{grp}_residual_vector_quantizer.layers.{i}.codebook"
        # HF:modeling_mimi.py:979-983
        tabs.append(w[p + ".embed_sum"] / w[p + ".cluster_usage"].clamp(min=1e-5)[:, None])
    out["codebooks"] = torch.cat(tabs, dim=0).contiguous()  # [Q*2048, 256], row q*2048 + tok
    psem = w["quantizer.semantic_residual_vector_quantizer.output_proj.weight"].squeeze(-1)
    pac = w["quantizer.acoustic_residual_vector_quantizer.output_proj.weight"].squeeze(-1)
    out["rvq_proj.w"] = torch.cat([psem, pac], dim=1).contiguous()  # [512, 256 sem | 256 ac]
    out["upsample.w"] = w["upsample.conv.weight"].squeeze(1).contiguous()  # [512, 4]
    for pre, name in (("tr", "decoder_transformer"), ("etr", "encoder_transformer")):
        if f"{name}.layers.0.mlp.fc1.weight" in w:
            _pack_transformer(out, w, pre, name, int(mc.num_hidden_layers))
    if "encoder.layers.0.conv.weight" in w:
        _pack_mimi_encoder(out, w, mc)
    out["sea.conv0.w"] = pack_conv1d(w["decoder.layers.0.conv.weight"])
    out["sea.conv0.b"] = w["decoder.layers.0.conv.bias"]
    li = 1
    for si, r in enumerate(mc.upsampling_ratios):
        li += 1
        out[f"sea.up{si}.w"], out[f"sea.up{si}.b"] = pack_convtr1d(w[f"decoder.layers.{li}.conv.weight"], w[f"decoder.layers.{li}.conv.bias"], int(r))
        li += 1
        p = f"decoder.layers.{li}.block"
        out[f"sea.res{si}.c1.w"] = pack_conv1d(w[p + ".1.conv.weight"])
        out[f"sea.res{si}.c1.b"] = w[p + ".1.conv.bias"]
        out[f"sea.res{si}.c2.w"] = pack_conv1d(w[p + ".3.conv.weight"])
        out[f"sea.res{si}.c2.b"] = w[p + ".3.conv.bias"]
        li += 1
    li += 1
    fw = w[f"decoder.layers.{li}.conv.weight"]  # [1, 64, 3]
    out["sea.final.w"] = fw[0].t().contiguous()  # [3, 64]
    out["sea.final.b"] = w[f"decoder.layers.{li}.conv.bias"].reshape(1)
    return {k: v.contiguous() for k, v in out.items()}


def _pack_mimi_encoder(out: Dict[str, torch.Tensor], w: Dict[str, torch.Tensor], mc: MimiDecoderConfig) -> None:
    """Encode side (HF:modeling_mimi.py MimiEncoder, downsample, RVQ input projections).  Strided convs keep the
    [Cout, tap*Cin + ci] layout: with stride r and k = 2r, output frame j contracts 2r consecutive channels-last rows."""
    out["enc.conv0.w"] = w["encoder.layers.0.conv.weight"].squeeze(1).contiguous()  # [64, 7]
    out["enc.conv0.b"] = w["encoder.layers.0.conv.bias"]
    li = 1
    for si, _r in enumerate(reversed(mc.upsampling_ratios)):
        p = f"encoder.layers.{li}.block"
        out[f"enc.res{si}.c1.w"] = pack_conv1d(w[p + ".1.conv.weight"])
        out[f"enc.res{si}.c1.b"] = w[p + ".1.conv.bias"]
        out[f"enc.res{si}.c2.w"] = pack_conv1d(w[p + ".3.conv.weight"])
        out[f"enc.res{si}.c2.b"] = w[p + ".3.conv.bias"]
        li += 2
        out[f"enc.down{si}.w"] = pack_conv1d(w[f"encoder.layers.{li}.conv.weight"])
        out[f"enc.down{si}.b"] = w[f"encoder.layers.{li}.conv.bias"]
        li += 1
    li += 1
    out["enc.final.w"] = pack_conv1d(w[f"encoder.layers.{li}.conv.weight"])
    out["enc.final.b"] = w[f"encoder.layers.{li}.conv.bias"]
    out["enc.ds.w"] = pack_conv1d(w["downsample.conv.weight"])  # [512, 4*512], no bias
    out["enc.inproj.sem.w"] = w["quantizer.semantic_residual_vector_quantizer.input_proj.weight"].squeeze(-1).contiguous()
    out["enc.inproj.ac.w"] = w["quantizer.acoustic_residual_vector_quantizer.input_proj.weight"].squeeze(-1).contiguous()
    # nearest code = argmax_e (r.e - |e|^2 / 2)
    out["enc.cb_bias"] = (-0.5 * (out["codebooks"].double() ** 2).sum(dim=1)).float()


def _pack_transformer(out: Dict[str, torch.Tensor], w: Dict[str, torch.Tensor], pre: str, name: str, n_layers: int) -> None:
    for li in range(n_layers):
        p = f"{name}.layers.{li}"
        out[f"{pre}.{li}.qkv.w"] = torch.cat([w[p + f".self_attn.{n}_proj.weight"] for n in ("q", "k", "v")], dim=0).contiguous()
        out[f"{pre}.{li}.o.w"] = w[p + ".self_attn.o_proj.weight"]
        out[f"{pre}.{li}.fc1.w"] = w[p + ".mlp.fc1.weight"]
        out[f"{pre}.{li}.fc2.w"] = w[p + ".mlp.fc2.weight"]
        out[f"{pre}.{li}.ln1.w"], out[f"{pre}.{li}.ln1.b"] = w[p + ".input_layernorm.weight"], w[p + ".input_layernorm.bias"]
        out[f"{pre}.{li}.ln2.w"], out[f"{pre}.{li}.ln2.b"] = w[p + ".post_attention_layernorm.weight"], w[p + ".post_attention_layernorm.bias"]
        out[f"{pre}.{li}.ls1"] = w[p + ".self_attn_layer_scale.scale"]
        out[f"{pre}.{li}.ls2"] = w[p + ".mlp_layer_scale.scale"]


def sinusoid_table(n: int, d: int) -> torch.Tensor:
    """Rows 0..n-1 of the reference's sinusoidal table (src/sopro/nn/embeddings.py:11-25), made on the
    host with the same torch ops so device code never evaluates sin/cos of positions."""
    import math

    pos = torch.arange(n, dtype=torch.float32).unsqueeze(1)
    div = torch.exp(torch.arange(0, d, 2, dtype=torch.float32) * (-math.log(10000.0) / d))
    pe = torch.zeros(n, d)
    pe[:, 0::2] = torch.sin(pos * div)
    pe[:, 1::2] = torch.cos(pos * div)
    return pe


def rope_tables(npos: int, dh: int, theta: float):
    """cos/sin [npos, dh/2] as HF computes them (HF:modeling_mimi.py:511-566)."""
    inv = 1.0 / (theta ** (torch.arange(0, dh, 2, dtype=torch.float32) / dh))
    fr = torch.arange(npos, dtype=torch.float32)[:, None] * inv[None, :]
    return fr.cos().contiguous(), fr.sin().contiguous()
