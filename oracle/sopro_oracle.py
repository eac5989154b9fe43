"""CPU oracle for the Sopro ``synthesize`` / ``stream`` hot path.  TEST INFRASTRUCTURE ONLY.

This file restates, as plain functions over a flat ``name -> tensor`` weight
dictionary, the arithmetic of the reference's hot path so that the HIP engine in
``sopro_amd/`` can be checked against it.  It is fp32 torch on the CPU (the path
is floating point: a torch fp32 restatement is the permitted form of the oracle).
Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline``
leg may import it; the product never does and fails loudly without its HIP library.

Pinning: the reference ships no tests and no golden vectors, so
the oracle is pinned against *the reference itself*: ``tests/golden/make_golden.py``
imports ``/root/reference/src/sopro`` and HuggingFace ``MimiModel`` in the build
container, runs them on seeded synthetic checkpoints and stores their outputs under
``tests/golden/*.npz``; ``tests/test_oracle_golden.py`` checks every function below
against those files.  On top of that ``tests/test_oracle_vs_hf_mimi.py`` re-runs the
installed third-party ``transformers`` Mimi decoder live (it is part of the image,
not of /root/reference).

Each function cites the reference lines it follows.  ``HF:`` = the third-party
``transformers/models/mimi/modeling_mimi.py`` (transformers 5.15.0 in this image;
the reference pins >=4.46, uv.lock 4.57.6 / 5.0.0).
"""
from __future__ import annotations

import math
from dataclasses import dataclass, field
from typing import Dict, Iterator, List, Optional, Sequence, Tuple

import torch
import torch.nn.functional as F

W = Dict[str, torch.Tensor]


def to_torch(weights: Dict[str, "object"]) -> W:
    out: W = {}
    for k, v in weights.items():
        t = v if isinstance(v, torch.Tensor) else torch.from_numpy(v)
        out[k] = t.float() if t.is_floating_point() else t
    return out


# =============================================================================
# building blocks
# =============================================================================
def rmsnorm(x: torch.Tensor, w: torch.Tensor, eps: float = 1e-6) -> torch.Tensor:
    """reference: src/sopro/nn/blocks.py:26-37"""
    x32 = x.float()
    var = x32.pow(2).mean(dim=-1, keepdim=True)
    return x32 * torch.rsqrt(var + eps) * w.float()


def sinusoid(positions: torch.Tensor, d_model: int) -> torch.Tensor:
    """reference: src/sopro/nn/embeddings.py:11-25 (table rows for ``positions``)."""
    pos = positions.float().unsqueeze(1)
    div = torch.exp(torch.arange(0, d_model, 2, dtype=torch.float32) * (-math.log(10000.0) / d_model))
    pe = torch.zeros(positions.numel(), d_model)
    pe[:, 0::2] = torch.sin(pos * div)
    pe[:, 1::2] = torch.cos(pos * div)
    return pe


def glu(x: torch.Tensor, w: W, p: str) -> torch.Tensor:
    """reference: src/sopro/nn/blocks.py:16-23"""
    y = F.linear(x, w[p + ".pro.weight"], w[p + ".pro.bias"])
    a, b = y.chunk(2, dim=-1)
    return a * torch.sigmoid(b)


def dwconv_full(h: torch.Tensor, wt: torch.Tensor, bias: torch.Tensor, dil: int, causal: bool) -> torch.Tensor:
    """Depthwise conv over time, [B,T,D] -> [B,T,D].
    reference: src/sopro/nn/blocks.py:63-74 (causal: left pad (k-1)d; else split total//2 left)."""
    k = int(wt.shape[-1])
    total = (k - 1) * dil
    left = total if causal else total // 2
    right = 0 if causal else total - left
    xt = F.pad(h.transpose(1, 2), (left, right))
    y = F.conv1d(xt, wt, bias, dilation=dil, groups=wt.shape[0])
    return y.transpose(1, 2)


def ssm_block(x: torch.Tensor, w: W, p: str, dil: int, causal: bool) -> torch.Tensor:
    """Full-sequence SSMLiteBlock.  reference: src/sopro/nn/blocks.py:143-148"""
    h = glu(rmsnorm(x, w[p + ".norm.weight"]), w, p + ".glu")
    h = dwconv_full(h, w[p + ".dw.dw.weight"], w[p + ".dw.dw.bias"], dil, causal)
    x = x + h
    f = rmsnorm(x, w[p + ".ff.0.weight"])
    f = F.linear(f, w[p + ".ff.1.weight"], w[p + ".ff.1.bias"])
    f = F.gelu(f)  # erf form (nn.GELU default), reference: src/sopro/nn/blocks.py:131
    f = F.linear(f, w[p + ".ff.3.weight"], w[p + ".ff.3.bias"])
    return x + f


def ssm_block_step(x: torch.Tensor, ring: torch.Tensor, w: W, p: str, dil: int) -> Tuple[torch.Tensor, torch.Tensor]:
    """One causal step, x [B,D], ring [B,L,D] (oldest row first).
    reference: src/sopro/nn/blocks.py:150-162 and :76-110 (shift, taps 0,d,2d.., weighted sum)."""
    h = glu(rmsnorm(x, w[p + ".norm.weight"]), w, p + ".glu")
    ring = torch.cat([ring[:, 1:, :], h.unsqueeze(1)], dim=1) if ring.size(1) > 1 else h.unsqueeze(1)
    wt = w[p + ".dw.dw.weight"].squeeze(1)  # [D,k]
    k = wt.shape[1]
    taps = ring[:, torch.arange(0, k * dil, dil), :]  # [B,k,D]
    y = (taps.transpose(1, 2) * wt.unsqueeze(0)).sum(dim=-1) + w[p + ".dw.dw.bias"]
    x = x + y
    f = rmsnorm(x, w[p + ".ff.0.weight"])
    f = F.gelu(F.linear(f, w[p + ".ff.1.weight"], w[p + ".ff.1.bias"]))
    f = F.linear(f, w[p + ".ff.3.weight"], w[p + ".ff.3.bias"])
    return x + f, ring


def _heads(t: torch.Tensor, h: int) -> torch.Tensor:
    b, n, d = t.shape
    return t.view(b, n, h, d // h).transpose(1, 2)


def _unheads(t: torch.Tensor) -> torch.Tensor:
    b, h, n, dh = t.shape
    return t.transpose(1, 2).reshape(b, n, h * dh)


def attention(q: torch.Tensor, k: torch.Tensor, v: torch.Tensor, keep: Optional[torch.Tensor]) -> torch.Tensor:
    """softmax(q k^T / sqrt(dh)) v in fp32; ``keep`` [B,S] bool (True = attend).
    Restates F.scaled_dot_product_attention as called at src/sopro/nn/text.py:118-126."""
    s = torch.matmul(q, k.transpose(-1, -2)) / math.sqrt(q.shape[-1])
    if keep is not None:
        s = s.masked_fill(~keep[:, None, None, :], float("-inf"))
    return torch.matmul(torch.softmax(s, dim=-1), v)


def xattn_kv(ctx: torch.Tensor, w: W, p: str, heads: int) -> Tuple[torch.Tensor, torch.Tensor]:
    """reference: src/sopro/nn/text.py:75-83 / src/sopro/nn/ref.py:44-52"""
    kv = rmsnorm(ctx, w[p + ".nkv.weight"])
    return _heads(F.linear(kv, w[p + ".k_proj.weight"]), heads), _heads(F.linear(kv, w[p + ".v_proj.weight"]), heads)


def text_xattn(x: torch.Tensor, k: torch.Tensor, v: torch.Tensor, keep: Optional[torch.Tensor], w: W, p: str) -> torch.Tensor:
    """Cached text cross-attention, x [B,T,D].  reference: src/sopro/nn/text.py:85-132.
    Rows whose keys are all masked attend to key 0 (:105-116)."""
    heads = k.shape[1]
    q = _heads(F.linear(rmsnorm(x, w[p + ".nq.weight"]), w[p + ".q_proj.weight"]), heads)
    if keep is not None:
        keep = keep.clone()
        bad = ~keep.any(dim=1)
        keep[bad, 0] = True
    a = attention(q, k, v, keep)
    a = torch.nan_to_num(a, nan=0.0, posinf=0.0, neginf=0.0)
    a = F.linear(_unheads(a), w[p + ".out_proj.weight"])
    return x + torch.tanh(w[p + ".gate"]) * a


def ref_xattn(x: torch.Tensor, k: torch.Tensor, v: torch.Tensor, w: W, p: str, gmax: float) -> torch.Tensor:
    """reference: src/sopro/nn/ref.py:54-108 (no padding mask on this path: PreparedReference
    stores key_padding_mask=None, src/sopro/model.py:163)."""
    heads = k.shape[1]
    q = _heads(F.linear(rmsnorm(x, w[p + ".nq.weight"]), w[p + ".q_proj.weight"]), heads)
    a = _unheads(torch.nan_to_num(attention(q, k, v, None), nan=0.0, posinf=0.0, neginf=0.0))

    def rms(t: torch.Tensor) -> torch.Tensor:  # reference: src/sopro/nn/ref.py:12-13
        return torch.sqrt(t.float().pow(2).mean(dim=-1, keepdim=True) + 1e-6)

    a = a * (rms(x) / rms(a)).clamp(0.0, 10.0)
    a = F.linear(a, w[p + ".out_proj.weight"])
    return x + (gmax * torch.tanh(w[p + ".gate"])) * a


# =============================================================================
# per-voice and per-utterance preparation
# =============================================================================
@dataclass
class OracleReference:
    """Field-for-field the reference's PreparedReference (src/sopro/model.py:45-50)."""

    ref_tokens_btq: torch.Tensor
    sv_ref: torch.Tensor
    ref_seq: torch.Tensor
    ref_kv_caches: List[Dict[str, Optional[torch.Tensor]]]


def token2sv(tok_btq: torch.Tensor, w: W, V: int) -> torch.Tensor:
    """reference: src/sopro/nn/speaker.py:37-61 with lengths == T (src/sopro/model.py:158-159)."""
    B, T, Q = tok_btq.shape
    idx = torch.arange(Q).view(1, 1, Q) * V + tok_btq.long()
    raw = w["token2sv.emb.weight"][idx]  # [B,T,Q,192]
    cw = torch.softmax(w["token2sv.cb_weights"], dim=0).view(1, 1, Q, 1)
    x = (raw * cw).sum(dim=2)
    h = x
    for i in (0, 3):
        h = F.gelu(dwconv_full(h, w[f"token2sv.enc.{i}.dw.weight"], w[f"token2sv.enc.{i}.dw.bias"], 1, False))
    # attentive statistics pooling, reference: src/sopro/nn/blocks.py:174-188
    logit = F.linear(torch.tanh(F.linear(h, w["token2sv.pool.attn.0.weight"], w["token2sv.pool.attn.0.bias"])),
                     w["token2sv.pool.attn.2.weight"], w["token2sv.pool.attn.2.bias"]).squeeze(-1)
    a = torch.softmax(logit, dim=1).unsqueeze(-1)
    mu = (h * a).sum(dim=1)
    std = torch.sqrt((a * (h - mu.unsqueeze(1)).pow(2)).sum(dim=1).clamp_min(1e-6))
    e = F.linear(torch.cat([mu, std], dim=-1), w["token2sv.proj.weight"], w["token2sv.proj.bias"])
    return F.normalize(e, dim=-1, eps=1e-6)


def encode_reference_seq(tok_btq: torch.Tensor, w: W, cfg) -> torch.Tensor:
    """reference: src/sopro/model.py:133-149"""
    B, T, Q = tok_btq.shape
    V = int(cfg.codebook_size)
    cw = torch.softmax(w["ref_cb_weights"].float(), dim=0)
    x = torch.zeros(B, T, int(cfg.d_model))
    for q in range(Q):
        x = x + cw[q] * w["cb_embed.emb.weight"][q * V + tok_btq[:, :, q].long()]
    for i in range(int(cfg.ref_enc_layers)):
        x = ssm_block(x, w, f"ref_enc_blocks.{i}", 1, False)
    return rmsnorm(x, w["ref_enc_norm.weight"])


def prepare_reference(ref_tokens_tq: torch.Tensor, w: W, cfg) -> OracleReference:
    """reference: src/sopro/model.py:151-170"""
    tok = ref_tokens_tq.unsqueeze(0).long()
    sv = token2sv(tok, w, int(cfg.codebook_size))
    seq = encode_reference_seq(tok, w, cfg)
    caches = []
    for i in range(int(cfg.ref_xattn_layers)):
        k, v = xattn_kv(seq, w, f"ref_xattn.blocks.{i}", int(cfg.ref_xattn_heads))
        caches.append({"k": k, "v": v, "key_padding_mask": None})
    return OracleReference(tok, sv, seq, caches)


def text_encoder(ids_bs: torch.Tensor, w: W, cfg) -> Tuple[torch.Tensor, torch.Tensor]:
    """reference: src/sopro/nn/text.py:29-44 with an all-true mask (src/sopro/model.py:186)."""
    d = int(cfg.d_model)
    x = w["text_enc.embed.emb.weight"][ids_bs.long()] + sinusoid(torch.arange(ids_bs.shape[1]), d).unsqueeze(0)
    for i in range(int(cfg.n_layers_text)):
        x = ssm_block(x, w, f"text_enc.layers.{i}", 1, False)
    x = rmsnorm(x, w["text_enc.norm.weight"])
    pooled = x.sum(dim=1) / (float(ids_bs.shape[1]) + 1e-6)
    return x, pooled


def speaker_film(base: torch.Tensor, sv: torch.Tensor, w: W, strength: float) -> torch.Tensor:
    """reference: src/sopro/nn/speaker.py:76-85 (nn.LayerNorm eps 1e-5)."""
    film = F.linear(F.gelu(F.linear(sv, w["spk_film.mlp.0.weight"], w["spk_film.mlp.0.bias"])),
                    w["spk_film.mlp.2.weight"], w["spk_film.mlp.2.bias"])
    gamma, beta = film.chunk(2, dim=-1)
    x = F.layer_norm(base, (base.shape[-1],), w["spk_film.norm.weight"], w["spk_film.norm.bias"], 1e-5)
    return x * (1 + strength * torch.tanh(gamma.unsqueeze(1))) + strength * torch.tanh(beta.unsqueeze(1))


def prepare_conditioning(ids_1d: torch.Tensor, ref: OracleReference, w: W, cfg, *, max_frames: int,
                         style_strength: float) -> Dict[str, torch.Tensor]:
    """reference: src/sopro/model.py:172-216"""
    d = int(cfg.d_model)
    txt_seq, txt_pool = text_encoder(ids_1d.view(1, -1), w, cfg)
    tar = int(max_frames) + 1
    base = txt_pool[:, None, :] + sinusoid(torch.arange(tar), d).unsqueeze(0)
    cond = speaker_film(base, ref.sv_ref.view(1, -1), w, float(style_strength))
    for i, c in enumerate(ref.ref_kv_caches):
        cond = ref_xattn(cond, c["k"], c["v"], w, f"ref_xattn.blocks.{i}", float(cfg.ref_xattn_gmax))
    cond = rmsnorm(cond, w["cond_norm.weight"])
    return {"txt_seq": txt_seq, "text_mask": torch.ones(1, ids_1d.numel(), dtype=torch.bool),
            "txt_pool": txt_pool, "sv_ref": ref.sv_ref.view(1, -1), "cond_ar": cond}


# =============================================================================
# autoregressive generator (codebook 0)
# =============================================================================
@dataclass
class ARState:
    rings: List[torch.Tensor]
    kv: Dict[int, Tuple[torch.Tensor, torch.Tensor]]
    keep: Optional[torch.Tensor]


def ar_init_state(B: int, txt_seq: torch.Tensor, text_mask: Optional[torch.Tensor], w: W, cfg) -> ARState:
    """reference: src/sopro/nn/generator.py:44-68, src/sopro/nn/blocks.py:50-61"""
    d = int(cfg.d_model)
    k = int(cfg.ar_kernel)
    rings = [torch.zeros(B, (k - 1) * dil + 1, d) for dil in cfg.ar_dilations]
    kv = {i: xattn_kv(txt_seq, w, f"ar.x_attns.{i}", 4) for i in cfg.ar_xattn_layers}
    return ARState(rings, kv, text_mask)


def ar_step(x_bd: torch.Tensor, st: ARState, w: W, cfg) -> torch.Tensor:
    """One frame: x [B,D] -> logits [B,V+1].  reference: src/sopro/nn/generator.py:98-130"""
    h = x_bd
    for i, dil in enumerate(cfg.ar_dilations):
        h, st.rings[i] = ssm_block_step(h, st.rings[i], w, f"ar.blocks.{i}", dil)
        if i in st.kv:
            k, v = st.kv[i]
            h = text_xattn(h.unsqueeze(1), k, v, st.keep, w, f"ar.x_attns.{i}").squeeze(1)
    return F.linear(rmsnorm(h, w["ar.norm.weight"]), w["ar.head.weight"], w["ar.head.bias"])


def ar_forward_teacher(x_btd: torch.Tensor, txt_seq: torch.Tensor, text_mask: Optional[torch.Tensor], w: W, cfg) -> torch.Tensor:
    """Parallel (teacher-forced) form.  reference: src/sopro/nn/generator.py:70-96"""
    h = x_btd
    for i, dil in enumerate(cfg.ar_dilations):
        h = ssm_block(h, w, f"ar.blocks.{i}", dil, True)
        if i in cfg.ar_xattn_layers:
            k, v = xattn_kv(txt_seq, w, f"ar.x_attns.{i}", 4)
            h = text_xattn(h, k, v, text_mask, w, f"ar.x_attns.{i}")
    return F.linear(rmsnorm(h, w["ar.norm.weight"]), w["ar.head.weight"], w["ar.head.bias"])


def repeated_tail(hist: Sequence[int], max_n: int = 16) -> bool:
    """reference: src/sopro/sampling.py:16-21"""
    L = len(hist)
    for n in range(3, min(max_n, L // 2) + 1):
        if list(hist[L - n:]) == list(hist[L - 2 * n: L - n]):
            return True
    return False


def penalised_logits(logits_v: torch.Tensor, history: Sequence[int], temperature: float, repetition_penalty: float) -> torch.Tensor:
    """Steps 1-3 of the sampler.  reference: src/sopro/sampling.py:33-50"""
    x = torch.nan_to_num(logits_v.float(), nan=-1e9, posinf=1e9, neginf=-1e9).clone()
    if temperature and temperature != 1.0:
        x = x / float(temperature)
    if repetition_penalty != 1.0 and len(history) > 0:
        ids = torch.tensor(sorted(set(history[-50:])), dtype=torch.long)
        vals = x[ids]
        x[ids] = torch.where(vals < 0, vals * repetition_penalty, vals / repetition_penalty)
    return x


def sampling_distribution(logits_v: torch.Tensor, history: Sequence[int], top_p: float, temperature: float,
                          top_k: int = 50, repetition_penalty: float = 1.1, eps: float = 1e-12
                          ) -> Tuple[torch.Tensor, torch.Tensor, Optional[int]]:
    """Everything of ``sample_token`` up to the random draw: returns (sorted_probs,
    sorted_token_ids, forced) where ``forced`` is the arg-max fallback token when the
    kept mass underflows.  reference: src/sopro/sampling.py:52-80"""
    x = penalised_logits(logits_v, history, temperature, repetition_penalty)
    probs = torch.nan_to_num(torch.softmax(x, dim=-1), nan=0.0, posinf=0.0, neginf=0.0)
    V = probs.numel()
    if top_k and top_k > 0:
        val, idx = torch.topk(probs, min(int(top_k), V))
        newp = torch.zeros_like(probs)
        newp[idx] = val
        s = newp.sum()
        if float(s) <= eps:
            return probs, torch.arange(V), int(torch.argmax(x))
        probs = newp / s
    sp, si = torch.sort(probs, descending=True)
    if top_p is not None and top_p < 1.0:
        remove = torch.cumsum(sp, dim=-1) > float(top_p)
        remove[1:] = remove[:-1].clone()
        remove[0] = False
        sp = sp.masked_fill(remove, 0.0)
    s = sp.sum()
    if float(s) <= eps:
        return sp, si, int(torch.argmax(x))
    return sp / s, si, None


def sample_token(logits_v: torch.Tensor, history: Sequence[int], top_p: float, temperature: float,
                 gen: Optional[torch.Generator] = None) -> int:
    """reference: src/sopro/sampling.py:24-93 with the policy constants of src/sopro/model.py:289-290."""
    sp, si, forced = sampling_distribution(logits_v, history, top_p, temperature)
    if forced is not None:
        return forced
    j = int(torch.multinomial(sp, 1, generator=gen))
    return int(si[j])


def ar_generate(prep: Dict[str, torch.Tensor], w: W, cfg, *, max_frames: int, top_p: float = 0.9,
                temperature: float = 1.05, anti_loop: bool = True, min_gen_frames: Optional[int] = None,
                gen: Optional[torch.Generator] = None, collect_logits: Optional[List[torch.Tensor]] = None,
                ) -> Iterator[Tuple[int, int, bool]]:
    """The sequential driver, batch 1.  reference: src/sopro/model.py:218-305"""
    cond = prep["cond_ar"]
    V = int(cfg.codebook_size)
    eos = V
    min_gen = int(min_gen_frames if min_gen_frames is not None else cfg.min_gen_frames)
    st = ar_init_state(1, prep["txt_seq"], prep["text_mask"], w, cfg)
    E = w["cb_embed.emb.weight"]
    hist: List[int] = []
    streak, last = 0, None
    for t in range(int(max_frames) + 1):
        prev = E[int(cfg.num_codebooks) * V] if t == 0 else E[hist[-1]]  # row 0*V+tok (src/sopro/nn/embeddings.py:51-55)
        x = cond[:, t, :] + prev.unsqueeze(0)
        p, tmp = top_p, temperature
        if anti_loop and (repeated_tail(hist, 16) or (last is not None and streak >= 8)):
            p, tmp = 0.85, 1.2
        logits = ar_step(x, st, w, cfg)[0]
        if collect_logits is not None:
            collect_logits.append(logits.clone())
        tok = sample_token(logits, hist, p, tmp, gen)
        hist.append(tok)
        streak = streak + 1 if (last is not None and tok == last) else 0
        last = tok
        is_eos = tok == eos
        yield t, tok, is_eos
        if is_eos and (t + 1) >= min_gen:
            break


# =============================================================================
# non-autoregressive refinement (codebooks 1..Q-1)
# =============================================================================
def nar_forward_stage(stage: str, sid: int, cond: torch.Tensor, prev: torch.Tensor, w: W, cfg) -> List[torch.Tensor]:
    """reference: src/sopro/nn/nar.py:89-116 and :13-32"""
    mix = torch.softmax(w[f"nar.mix.{stage}"], dim=0)
    x = mix[0] * cond + mix[1] * prev
    sv = w["nar.stage_emb.weight"][sid].unsqueeze(0)
    gb = F.linear(F.gelu(F.linear(sv, w["nar.adapter.mlp.0.weight"], w["nar.adapter.mlp.0.bias"])),
                  w["nar.adapter.mlp.2.weight"], w["nar.adapter.mlp.2.bias"])
    g, b = gb.chunk(2, dim=-1)
    x = rmsnorm(x, w["nar.adapter.norm.weight"]) * (1 + torch.tanh(g.unsqueeze(1))) + torch.tanh(b.unsqueeze(1))
    for i, dil in enumerate(cfg.nar_dilations):
        x = ssm_block(x, w, f"nar.blocks.{i}", dil, False)
    z = F.linear(rmsnorm(x, w["nar.norm.weight"]), w["nar.pre.weight"], w["nar.pre.bias"])
    outs = []
    n_heads = len(cfg.stage_codebooks()[stage])
    for i in range(n_heads):
        hb = w[f"nar.head_id_emb.{stage}.weight"][i].view(1, 1, -1)
        outs.append(F.linear(z + hb, w[f"nar.heads.{stage}.{i}.weight"], w[f"nar.heads.{stage}.{i}.bias"]))
    return outs


def nar_refine(cond_btd: torch.Tensor, rvq1_bt: torch.Tensor, w: W, cfg, *,
               collect_logits: Optional[Dict[int, torch.Tensor]] = None) -> torch.Tensor:
    """reference: src/sopro/model.py:307-347 (+ sum_embed_subset, src/sopro/nn/embeddings.py:77-112)"""
    B, T, _ = cond_btd.shape
    Q, V = int(cfg.num_codebooks), int(cfg.codebook_size)
    out = torch.zeros(B, T, Q, dtype=torch.long)
    out[:, :, 0] = rvq1_bt
    known = [0]
    E = w["cb_embed.emb.weight"]
    sc = cfg.stage_codebooks()
    for sid, stage in enumerate(cfg.stage_order()):
        cbs = sc[stage]
        cw = torch.softmax(w["nar_prev_cb_weights"][torch.tensor(known)].float(), dim=0)
        prev = torch.zeros(B, T, int(cfg.d_model))
        for j, cb in enumerate(known):
            prev = prev + cw[j] * E[cb * V + out[:, :, cb]]
        logits = nar_forward_stage(stage, sid, cond_btd, prev, w, cfg)
        for j, cb in enumerate(cbs):
            out[:, :, cb] = logits[j].argmax(dim=-1)
            if collect_logits is not None:
                collect_logits[cb] = logits[j]
        known = known + cbs
    return out


def nar_audit(cond_btd: torch.Tensor, tokens_btq: torch.Tensor, w: W, cfg) -> Tuple[int, float]:
    """Teacher-forced audit of a refined token matrix (test infrastructure for near-tie arg-maxes): every stage is run on
    the embeddings of the GIVEN tokens of the codebooks decided before it (so one flipped near-tie does not make every later
    decision look wrong), and each given token is compared with this oracle's arg-max.  Returns (number of positions where
    they differ, the largest logit gap ``max - logit[given]`` among those): a correct engine differs only where the gap is
    round-off sized.  Same arithmetic as nar_refine above (src/sopro/model.py:307-347)."""
    B, T, Q = tokens_btq.shape
    V = int(cfg.codebook_size)
    known = [0]
    E = w["cb_embed.emb.weight"]
    sc = cfg.stage_codebooks()
    n_off, worst = 0, 0.0
    for sid, stage in enumerate(cfg.stage_order()):
        cbs = sc[stage]
        cw = torch.softmax(w["nar_prev_cb_weights"][torch.tensor(known)].float(), dim=0)
        prev = torch.zeros(B, T, int(cfg.d_model))
        for j, cb in enumerate(known):
            prev = prev + cw[j] * E[cb * V + tokens_btq[:, :, cb]]
        logits = nar_forward_stage(stage, sid, cond_btd, prev, w, cfg)
        for j, cb in enumerate(cbs):
            given = tokens_btq[:, :, cb]
            gap = logits[j].max(dim=-1).values - logits[j].gather(-1, given.unsqueeze(-1)).squeeze(-1)
            off = logits[j].argmax(dim=-1) != given
            n_off += int(off.sum())
            if bool(off.any()):
                worst = max(worst, float(gap[off].max()))
        known = known + cbs
    return n_off, worst


def ar_audit_greedy(prep: Dict[str, torch.Tensor], tokens: Sequence[int], w: W, cfg, *, temperature: float = 1.0,
                    repetition_penalty: float = 1.1) -> Tuple[int, float]:
    """Teacher-forced audit of a greedy codebook-0 token list: the loop of ar_generate fed with the GIVEN history; returns
    (steps where the given token is not this oracle's arg-max of the penalised logits, largest gap among those)."""
    cond = prep["cond_ar"]
    V = int(cfg.codebook_size)
    st = ar_init_state(1, prep["txt_seq"], prep["text_mask"], w, cfg)
    E = w["cb_embed.emb.weight"]
    hist: List[int] = []
    n_off, worst = 0, 0.0
    for t, tok in enumerate(tokens):
        prev = E[int(cfg.num_codebooks) * V] if t == 0 else E[hist[-1]]
        xs = penalised_logits(ar_step(cond[:, t, :] + prev.unsqueeze(0), st, w, cfg)[0], hist, temperature, repetition_penalty)
        if int(xs.argmax()) != int(tok):
            n_off += 1
            worst = max(worst, float(xs.max() - xs[int(tok)]))
        hist.append(int(tok))
    return n_off, worst


def generate_tokens(ids_1d: torch.Tensor, ref: OracleReference, w: W, cfg, *, max_frames: int, top_p: float = 0.9,
                    temperature: float = 1.05, anti_loop: bool = True, style_strength: float = 1.0,
                    min_gen_frames: Optional[int] = None, gen: Optional[torch.Generator] = None) -> torch.Tensor:
    """reference: src/sopro/model.py:349-401 (cut at the FIRST EOS, even one sampled before min_gen_frames)."""
    prep = prepare_conditioning(ids_1d, ref, w, cfg, max_frames=max_frames, style_strength=style_strength)
    hist: List[int] = []
    for _t, tok, is_eos in ar_generate(prep, w, cfg, max_frames=max_frames, top_p=top_p, temperature=temperature,
                                       anti_loop=anti_loop, min_gen_frames=min_gen_frames, gen=gen):
        hist.append(tok)
        if is_eos:
            break
    eos = int(cfg.codebook_size)
    T = hist.index(eos) if eos in hist else len(hist)
    if T <= 0:
        return torch.zeros(0, int(cfg.num_codebooks), dtype=torch.long)
    rvq1 = torch.tensor(hist[:T], dtype=torch.long).unsqueeze(0)
    return nar_refine(prep["cond_ar"][:, :T, :], rvq1, w, cfg).squeeze(0)


# =============================================================================
# Mimi codec, decode side (third-party arithmetic, HF transformers)
# =============================================================================
def mimi_codebooks(mw: W, mc) -> List[torch.Tensor]:
    """embed = embed_sum / clamp(cluster_usage, 1e-5).  HF:modeling_mimi.py:979-983"""
    out = []
    for q in range(int(mc.num_quantizers)):
        grp, i = ("semantic", q) if q < mc.num_semantic_quantizers else ("acoustic", q - mc.num_semantic_quantizers)
        p = f"quantizer.{grp}_residual_vector_quantizer.layers.{i}.codebook"
        out.append(mw[p + ".embed_sum"] / mw[p + ".cluster_usage"].clamp(min=1e-5)[:, None])
    return out


def mimi_quantizer_decode(codes_bqt: torch.Tensor, mw: W, mc) -> torch.Tensor:
    """[B,Q,T] -> [B,512,T].  HF:modeling_mimi.py:1128-1137, 1070-1081"""
    cbs = mimi_codebooks(mw, mc)
    ns = int(mc.num_semantic_quantizers)
    sem = sum(cbs[q][codes_bqt[:, q]] for q in range(ns)).transpose(1, 2)
    out = F.conv1d(sem, mw["quantizer.semantic_residual_vector_quantizer.output_proj.weight"])
    if codes_bqt.shape[1] > ns:
        ac = sum(cbs[q][codes_bqt[:, q]] for q in range(ns, codes_bqt.shape[1])).transpose(1, 2)
        out = out + F.conv1d(ac, mw["quantizer.acoustic_residual_vector_quantizer.output_proj.weight"])
    return out


def causal_conv1d(x: torch.Tensor, wt: torch.Tensor, b: Optional[torch.Tensor]) -> torch.Tensor:
    """Stride-1 causal conv, left zero pad k-1.  HF:modeling_mimi.py:327-347 (extra padding is 0 at stride 1)."""
    return F.conv1d(F.pad(x, (int(wt.shape[-1]) - 1, 0)), wt, b)


def causal_convtr1d(x: torch.Tensor, wt: torch.Tensor, b: Optional[torch.Tensor], stride: int, groups: int = 1) -> torch.Tensor:
    """ConvTranspose1d, trim k-stride samples on the right.  HF:modeling_mimi.py:379-387,399-405"""
    y = F.conv_transpose1d(x, wt, b, stride=stride, groups=groups)
    return y[..., : y.shape[-1] - (int(wt.shape[-1]) - stride)]


def rope_cos_sin(pos: torch.Tensor, dh: int, theta: float) -> Tuple[torch.Tensor, torch.Tensor]:
    """HF:modeling_mimi.py:511-566"""
    inv = 1.0 / (theta ** (torch.arange(0, dh, 2, dtype=torch.float32) / dh))
    fr = pos.float()[:, None] * inv[None, :]
    emb = torch.cat([fr, fr], dim=-1)
    return emb.cos(), emb.sin()


def _rot_half(x: torch.Tensor) -> torch.Tensor:
    h = x.shape[-1] // 2
    return torch.cat([-x[..., h:], x[..., :h]], dim=-1)


@dataclass
class MimiKV:
    """Decoder-transformer cache as the installed transformers keeps it on the reference's
    streaming path: append-only."""

    k: List[Optional[torch.Tensor]] = field(default_factory=list)
    v: List[Optional[torch.Tensor]] = field(default_factory=list)
    seen: int = 0
    evict: bool = True  # False once rebuilt through the legacy-cache API (plain layers: nothing is dropped, see decode_step)


def mimi_transformer(x_bnc: torch.Tensor, mw: W, mc, cache: Optional[MimiKV] = None, prefix: str = "decoder_transformer") -> torch.Tensor:
    """8 pre-LN layers, RoPE, causal sliding window, LayerScale.  HF:modeling_mimi.py:729-928"""
    B, N, C = x_bnc.shape
    H, dh = int(mc.num_attention_heads), int(mc.head_dim)
    past = cache.seen if cache is not None else 0
    pos = torch.arange(N) + past
    cos, sin = rope_cos_sin(pos, dh, float(mc.rope_theta))
    win = int(mc.sliding_window)
    h = x_bnc
    for li in range(int(mc.num_hidden_layers)):
        p = f"{prefix}.layers.{li}"
        y = F.layer_norm(h, (C,), mw[p + ".input_layernorm.weight"], mw[p + ".input_layernorm.bias"], float(mc.norm_eps))
        q = _heads(F.linear(y, mw[p + ".self_attn.q_proj.weight"]), H)
        k = _heads(F.linear(y, mw[p + ".self_attn.k_proj.weight"]), H)
        v = _heads(F.linear(y, mw[p + ".self_attn.v_proj.weight"]), H)
        q = q * cos + _rot_half(q) * sin
        k = k * cos + _rot_half(k) * sin
        kpos = pos
        if cache is not None:
            if len(cache.k) <= li:
                cache.k.append(None)
                cache.v.append(None)
            if cache.k[li] is not None:
                k = torch.cat([cache.k[li], k], dim=2)
                v = torch.cat([cache.v[li], v], dim=2)
            # DynamicSlidingWindowLayer keeps the last (window-1) positions for the next call
            cache.k[li], cache.v[li] = (k[:, :, -(win - 1):], v[:, :, -(win - 1):]) if cache.evict else (k, v)
            kpos = torch.arange(past + N - k.shape[2], past + N)
        s = torch.matmul(q, k.transpose(-1, -2)) / math.sqrt(dh)
        vis = (kpos[None, :] <= pos[:, None]) & (kpos[None, :] > pos[:, None] - win)
        s = s.masked_fill(~vis[None, None], float("-inf"))
        a = _unheads(torch.matmul(torch.softmax(s, dim=-1), v))
        a = F.linear(a, mw[p + ".self_attn.o_proj.weight"])
        h = h + mw[p + ".self_attn_layer_scale.scale"] * a
        y = F.layer_norm(h, (C,), mw[p + ".post_attention_layernorm.weight"], mw[p + ".post_attention_layernorm.bias"], float(mc.norm_eps))
        y = F.linear(F.gelu(F.linear(y, mw[p + ".mlp.fc1.weight"])), mw[p + ".mlp.fc2.weight"])
        h = h + mw[p + ".mlp_layer_scale.scale"] * y
    if cache is not None:
        cache.seen = past + N
    return h


def seanet_decoder(x_bct: torch.Tensor, mw: W, mc, taps: Optional[Dict[str, torch.Tensor]] = None) -> torch.Tensor:
    """HF:modeling_mimi.py:931-961 (+ MimiResnetBlock :408-447)"""
    h = causal_conv1d(x_bct, mw["decoder.layers.0.conv.weight"], mw["decoder.layers.0.conv.bias"])
    li = 1
    for r in mc.upsampling_ratios:
        li += 1
        h = causal_convtr1d(F.elu(h), mw[f"decoder.layers.{li}.conv.weight"], mw[f"decoder.layers.{li}.conv.bias"], int(r))
        li += 1
        p = f"decoder.layers.{li}.block"
        y = causal_conv1d(F.elu(h), mw[p + ".1.conv.weight"], mw[p + ".1.conv.bias"])
        y = causal_conv1d(F.elu(y), mw[p + ".3.conv.weight"], mw[p + ".3.conv.bias"])
        h = h + y
        if taps is not None:
            taps[f"seanet_stage_{r}"] = h
        li += 1
    li += 1
    return causal_conv1d(F.elu(h), mw[f"decoder.layers.{li}.conv.weight"], mw[f"decoder.layers.{li}.conv.bias"])


def mimi_decode(codes_bqt: torch.Tensor, mw: W, mc, cache: Optional[MimiKV] = None,
                taps: Optional[Dict[str, torch.Tensor]] = None) -> torch.Tensor:
    """[B,Q,T] int -> [B,1,T*1920].  HF:modeling_mimi.py:1388-1406"""
    emb = mimi_quantizer_decode(codes_bqt.long(), mw, mc)
    up = causal_convtr1d(emb, mw["upsample.conv.weight"], None, int(mc.upsample_stride), groups=emb.shape[1])
    tr = mimi_transformer(up.transpose(1, 2), mw, mc, cache).transpose(1, 2)
    if taps is not None:
        taps["rvq"], taps["upsample"], taps["transformer"] = emb, up, tr
    return seanet_decoder(tr, mw, mc, taps)


# -----------------------------------------------------------------------------
# Mimi codec, encode side (reference audio -> tokens)
# -----------------------------------------------------------------------------
def causal_conv1d_strided(x: torch.Tensor, wt: torch.Tensor, b: Optional[torch.Tensor], stride: int, pad_mode: str = "constant") -> torch.Tensor:
    """MimiConv1d.forward, causal: left pad k - stride, right pad up to a whole frame.  HF:modeling_mimi.py:300-347"""
    k = int(wt.shape[-1])
    L = int(x.shape[-1])
    pt = k - stride
    n_frames = math.ceil((L - k + pt) / stride + 1) - 1
    extra = n_frames * stride + k - pt - L
    return F.conv1d(F.pad(x, (pt, extra), mode=pad_mode), wt, b, stride=stride)


def seanet_encoder(x_b1n: torch.Tensor, mw: W, mc) -> torch.Tensor:
    """HF:modeling_mimi.py MimiEncoder (conv k7, then per ratio [residual block, ELU, strided conv], ELU, conv k3)."""
    h = causal_conv1d(x_b1n, mw["encoder.layers.0.conv.weight"], mw["encoder.layers.0.conv.bias"])
    li = 1
    for r in reversed(mc.upsampling_ratios):
        p = f"encoder.layers.{li}.block"
        y = causal_conv1d(F.elu(h), mw[p + ".1.conv.weight"], mw[p + ".1.conv.bias"])
        y = causal_conv1d(F.elu(y), mw[p + ".3.conv.weight"], mw[p + ".3.conv.bias"])
        h = h + y
        li += 2
        h = causal_conv1d_strided(F.elu(h), mw[f"encoder.layers.{li}.conv.weight"], mw[f"encoder.layers.{li}.conv.bias"], int(r))
        li += 1
    li += 1
    return causal_conv1d(F.elu(h), mw[f"encoder.layers.{li}.conv.weight"], mw[f"encoder.layers.{li}.conv.bias"])


def rvq_encode(emb_bct: torch.Tensor, mw: W, mc) -> torch.Tensor:
    """Split residual VQ encode -> codes [B, Q, T].  HF:modeling_mimi.py MimiSplitResidualVectorQuantizer.encode,
    MimiEuclideanCodebook.quantize (cdist + argmin)."""
    cbs = mimi_codebooks(mw, mc)
    ns = int(mc.num_semantic_quantizers)
    out = []
    for group, q0, q1 in (("semantic", 0, ns), ("acoustic", ns, int(mc.num_quantizers))):
        res = F.conv1d(emb_bct, mw[f"quantizer.{group}_residual_vector_quantizer.input_proj.weight"])
        for q in range(q0, q1):
            r = res.permute(0, 2, 1)  # [B, T, D]
            d = torch.cdist(r.reshape(1, -1, r.shape[-1]).float(), cbs[q][None].float(), p=2)[0]
            idx = d.argmin(dim=-1).view(r.shape[0], r.shape[1])
            res = res - cbs[q][idx].permute(0, 2, 1)
            out.append(idx)
    return torch.stack(out, dim=1)


def mimi_encode(wav_b1n: torch.Tensor, mw: W, mc, taps: Optional[Dict[str, torch.Tensor]] = None) -> torch.Tensor:
    """waveform [B, 1, N] -> codes [B, Q, T].  HF:modeling_mimi.py MimiModel._encode_frame"""
    emb = seanet_encoder(wav_b1n, mw, mc)
    tr = mimi_transformer(emb.transpose(1, 2), mw, mc, None, prefix="encoder_transformer").transpose(1, 2)
    ds = causal_conv1d_strided(tr, mw["downsample.conv.weight"], None, int(mc.upsample_stride), pad_mode="replicate")
    if taps is not None:
        taps["enc_seanet"], taps["enc_transformer"], taps["enc_downsample"] = emb, tr, ds
    return rvq_encode(ds, mw, mc)


def trim_silence_energy(wav: torch.Tensor, sr: int, frame_ms: float = 25.0, hop_ms: float = 10.0, thresh_db_floor: float = -40.0,
                        prepad_ms: float = 30.0, postpad_ms: float = 30.0, min_keep_sec: float = 0.5) -> torch.Tensor:
    """Energy-threshold silence trim of a mono [N] waveform.  reference: src/sopro/audio.py:30-86"""
    T = int(wav.shape[-1])
    frame_len, hop = max(1, int(sr * frame_ms / 1000.0)), max(1, int(sr * hop_ms / 1000.0))
    if T == 0 or T < int(sr * 0.1) or T < frame_len:
        return wav
    energy = wav.unfold(-1, frame_len, hop).pow(2).mean(dim=-1)
    energy_db = 10.0 * torch.log10(energy + 1e-10)
    thresh_db = max(float(energy_db.max()) + thresh_db_floor, thresh_db_floor)
    idx = torch.nonzero(energy_db > thresh_db)
    if idx.numel() == 0:
        return wav
    start = max(0, int(idx[0, 0]) * hop - int(sr * prepad_ms / 1000.0))
    end = min(T, int(idx[-1, 0]) * hop + frame_len + int(sr * postpad_ms / 1000.0))
    if end <= start or (end - start) < int(min_keep_sec * sr):
        return wav
    return wav[start:end]


def center_crop_audio(wav: torch.Tensor, win_samples: int) -> torch.Tensor:
    """reference: src/sopro/audio.py:148-155"""
    T = int(wav.shape[-1])
    if win_samples <= 0 or T <= win_samples:
        return wav
    s = (T - win_samples) // 2
    return wav[..., s:s + win_samples]


def sinc_resample(wav_n: torch.Tensor, sr_in: int, sr_out: int, lowpass_filter_width: int = 6, rolloff: float = 0.99) -> torch.Tensor:
    """reference: src/sopro/audio.py:113-123 calls ``torchaudio.functional.resample(wav, sr_in, sr_out)``.  torchaudio is
    NOT in this image, so this restates its published algorithm (functional.py ``_get_sinc_resample_kernel`` +
    ``_apply_sinc_resample_kernel``, Hann-windowed sinc, defaults width 6 / rolloff 0.99) -- PARITY UNPINNED for this
    one function: no torchaudio output was available to check it against."""
    if int(sr_in) == int(sr_out):
        return wav_n
    g = math.gcd(int(sr_in), int(sr_out))
    orig, new = int(sr_in) // g, int(sr_out) // g
    base = min(orig, new) * rolloff
    width = math.ceil(lowpass_filter_width * orig / base)
    idx = torch.arange(-width, width + orig, dtype=torch.float64)[None, None] / orig
    t = torch.arange(0, -new, -1, dtype=torch.float64)[:, None, None] / new + idx
    t = (t * base).clamp_(-lowpass_filter_width, lowpass_filter_width)
    window = torch.cos(t * math.pi / lowpass_filter_width / 2) ** 2
    t = t * math.pi
    kern = torch.where(t == 0, torch.ones_like(t), t.sin() / t) * window * (base / orig)
    n = int(wav_n.shape[-1])
    x = F.pad(wav_n.view(1, 1, n).float(), (width, width + orig))
    y = F.conv1d(x, kern.float(), stride=orig).transpose(1, 2).reshape(-1)
    return y[: math.ceil(new * n / orig)]


def encode_audio(wav_n: torch.Tensor, sr: int, mw: W, mc, crop_seconds: Optional[float] = None) -> torch.Tensor:
    """mono waveform -> codes [T, Q].  reference: src/sopro/codec/mimi.py:42-63 (after the file has been read)"""
    wav = trim_silence_energy(wav_n, sr)
    sr_t = int(mc.sampling_rate)
    wav = sinc_resample(wav, sr, sr_t)
    if crop_seconds is not None and crop_seconds > 0:
        hop = int(round(sr_t / float(mc.frame_rate)))
        wav = center_crop_audio(wav, max(1, int(round(crop_seconds * float(mc.frame_rate)))) * hop)
    return mimi_encode(wav.view(1, 1, -1), mw, mc)[0].permute(1, 0).contiguous()


def decode_full(tokens_tq: torch.Tensor, mw: W, mc) -> torch.Tensor:
    """reference: src/sopro/codec/mimi.py:65-72 -> [1,1,T*1920]"""
    return mimi_decode(tokens_tq.permute(1, 0).unsqueeze(0).contiguous(), mw, mc)


@dataclass
class StreamDecodeState:
    """reference: src/sopro/codec/mimi.py:75-80"""

    cache: Optional[MimiKV] = None
    frames_seen: int = 0
    samples_emitted: int = 0
    tail_codes_tq: Optional[torch.Tensor] = None


def decode_step(codes_chunk_tq: torch.Tensor, st: StreamDecodeState, mw: W, mc, overlap_frames: int = 2, trim: str = "none"
                ) -> Tuple[torch.Tensor, StreamDecodeState]:
    """reference: src/sopro/codec/mimi.py:115-181.  ``trim="none"``: as it behaves with the installed transformers 5.x
    (``drop_cache_tail`` finds no legacy-cache API and trims nothing: SURVEY.md Appendix C).  ``trim="legacy"``: the
    legacy branch of ``drop_cache_tail`` (:92-103, transformers 4.57.6 API): the last ``ov`` cached positions of every
    layer are dropped, the cache is rebuilt as plain (k, v) layers - which never evict - and the positions of the next
    call continue from the trimmed length (quirk Q6: ``ov`` frames are 2*ov positions, only ov are dropped)."""
    if trim not in ("none", "legacy"):
        raise ValueError("trim must be 'none' or 'legacy'")
    hop = int(mc.frame_samples)
    n_new = int(codes_chunk_tq.shape[0])
    if n_new == 0:
        return torch.zeros(1, 0), st
    ov = 0
    codes_in = codes_chunk_tq
    if overlap_frames > 0 and st.tail_codes_tq is not None and st.tail_codes_tq.numel() > 0:
        ov = min(int(overlap_frames), int(st.tail_codes_tq.shape[0]))
        codes_in = torch.cat([st.tail_codes_tq[-ov:], codes_chunk_tq], dim=0)
    if st.cache is None:
        st.cache = MimiKV()
    if trim == "legacy" and ov > 0 and st.cache.k and st.cache.k[0] is not None:
        c = st.cache
        c.k = [t[:, :, : max(0, t.shape[2] - ov)] for t in c.k]
        c.v = [t[:, :, : max(0, t.shape[2] - ov)] for t in c.v]
        c.seen, c.evict = int(c.k[0].shape[2]), False
    wav = mimi_decode(codes_in.permute(1, 0).unsqueeze(0).contiguous(), mw, mc, st.cache).reshape(1, -1)
    wav = wav[:, : (ov + n_new) * hop][:, ov * hop:]
    st.frames_seen += n_new
    st.samples_emitted += int(wav.shape[1])
    keep = min(int(overlap_frames), int(codes_in.shape[0]))
    st.tail_codes_tq = codes_in[-keep:].clone() if overlap_frames > 0 else None
    return wav, st


# =============================================================================
# public-API level drivers
# =============================================================================
def synthesize(ids_1d: torch.Tensor, ref: OracleReference, w: W, mw: W, cfg, mc, **kw) -> torch.Tensor:
    """reference: src/sopro/model.py:531-575 (text already tokenised)."""
    kw.setdefault("style_strength", float(cfg.style_strength))
    tokens = generate_tokens(ids_1d, ref, w, cfg, **kw)
    return decode_full(tokens, mw, mc)


def stream(ids_1d: torch.Tensor, ref: OracleReference, w: W, mw: W, cfg, mc, *, max_frames: int = 400,
           top_p: float = 0.9, temperature: float = 1.05, anti_loop: bool = True, style_strength: Optional[float] = None,
           chunk_frames: int = 6, nar_context_frames: Optional[int] = None, min_gen_frames: Optional[int] = None,
           gen: Optional[torch.Generator] = None, trim: str = "none") -> Iterator[torch.Tensor]:
    """reference: src/sopro/streaming.py:24-152 (stops at the first EOS, :114-115)."""
    prep = prepare_conditioning(ids_1d, ref, w, cfg, max_frames=max_frames,
                                style_strength=float(style_strength if style_strength is not None else cfg.style_strength))
    nar_ctx = int(nar_context_frames if nar_context_frames is not None else cfg.rf_nar())
    hist: List[int] = []
    emitted = 0
    st = StreamDecodeState()

    def refine_and_emit(end: int) -> Optional[torch.Tensor]:
        nonlocal emitted, st
        if end <= emitted:
            return None
        ws = max(0, emitted - nar_ctx)
        toks = nar_refine(prep["cond_ar"][:, ws:end, :], torch.tensor(hist[ws:end]).unsqueeze(0), w, cfg).squeeze(0)
        wav, st = decode_step(toks[emitted - ws:, :], st, mw, mc, trim=trim)
        emitted = end
        return wav if wav.numel() > 0 else None

    for _t, tok, is_eos in ar_generate(prep, w, cfg, max_frames=max_frames, top_p=top_p, temperature=temperature,
                                       anti_loop=anti_loop, min_gen_frames=min_gen_frames, gen=gen):
        if is_eos:
            break
        hist.append(tok)
        if len(hist) % int(chunk_frames) == 0:
            wav = refine_and_emit(len(hist))
            if wav is not None:
                yield wav
    if emitted < len(hist):
        wav = refine_and_emit(len(hist))
        if wav is not None:
            yield wav
