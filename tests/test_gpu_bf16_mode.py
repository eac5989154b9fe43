"""The engine's bf16 mode (BASELINE.json configs[1] / SURVEY.md 8d config 2: bf16 weights and activations into the matrix
cores, fp32 accumulators / norms / softmax): ``sopro_gemm_bf16x1`` against a torch product of bf16-rounded operands, and
the mode's end-to-end quality against the fp32 reference fixtures - reported, with floors that catch a broken path.
This is NOT the parity configuration: precision="f32" is, and every other test file checks that one."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

from conftest import FakeTok, golden
from oracle import sopro_oracle as O
from sopro_amd import hip, pack

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def rnd(*shape, seed=0, scale=1.0):
    g = torch.Generator().manual_seed(seed)
    return torch.randn(*shape, generator=g) * scale


def dev(t):
    return t.to(DEV).contiguous()


def bf(t):
    return t.to(torch.bfloat16).to(torch.float32)


def close(a, b, atol, what=""):
    err = float((a.detach().float().cpu() - b.detach().float().cpu()).abs().max())
    assert err <= atol, f"{what}: max abs err {err:.3e} > {atol:.1e}"


@pytest.mark.parametrize("M,N,K", [(300, 200, 96), (129, 64, 64), (70, 33, 192), (257, 2048, 256), (6400, 384, 1536), (12800, 512, 512),
                                   (6, 768, 384), (12, 2048, 2048), (1, 64, 64), (33, 96, 320)])
def test_gemm_bf16x1_is_the_product_of_the_rounded_operands(M, N, K):
    """One pass: exactly sum_k bf16(a) * bf16(w) in fp32 accumulation (only the summation order differs from torch's)."""
    A, W, b = rnd(M, K, seed=1), rnd(N, K, seed=2, scale=K ** -0.5), rnd(N, seed=3)
    C = torch.full((M, N), float("nan"), device=DEV)
    hip.gemm(dev(A), hip.pack_w_bf16x1(dev(W)), C, M=M, N=N, K=K, bias=dev(b))
    torch.cuda.synchronize()
    ref = bf(A).double() @ bf(W).double().t() + b.double()
    bound = (bf(A).double().abs() @ bf(W).double().abs().t() + b.double().abs()) * 2.0 ** -21
    err = (C.cpu().double() - ref).abs()
    assert bool((err <= bound).all()), f"{M}x{N}x{K}: worst {float((err / bound).max()):.2f} of the bound"
    # and it is a bf16-class approximation of the fp32 product
    full = A.double() @ W.double().t() + b.double()
    assert float((C.cpu().double() - full).abs().max()) < 2.0 ** -7 * float((A.abs() @ W.abs().t()).max())


def test_gemm_bf16x1_epilogues_prologues_norm_and_row_windows():
    M, N, K = 200, 256, 128
    A, W, b = rnd(M, K, seed=4), rnd(N, K, seed=5, scale=K ** -0.5), rnd(N, seed=6)
    R, pv = rnd(M, N, seed=7), rnd(K, seed=9)
    Wp = hip.pack_w_bf16x1(dev(W))
    ref = bf(A) @ bf(W).t() + b
    C = torch.empty(M, N, device=DEV)
    hip.gemm(dev(A), Wp, C, M=M, N=N, K=K, bias=dev(b), epilogue=hip.EPI_GELU)
    close(C, F.gelu(ref), 3e-5, "gelu")
    hip.gemm(dev(A), Wp, C, M=M, N=N, K=K, bias=dev(b), epilogue=hip.EPI_RES, R=dev(R))
    close(C, R + ref, 5e-5, "res")
    hip.gemm(dev(A), Wp, C, M=M, N=N, K=K, bias=dev(b), prologue=hip.PRO_ADDVEC, pro_vec=dev(pv))
    close(C, bf(A + pv) @ bf(W).t() + b, 5e-5, "addvec prologue")
    hip.gemm(dev(A), Wp, C, M=M, N=N, K=K, bias=dev(b), prologue=hip.PRO_ELU)
    close(C, bf(F.elu(A)) @ bf(W).t() + b, 5e-5, "elu prologue")
    C2 = torch.empty(M, N, device=DEV)
    hip.gemm(dev(A), Wp, C, M=M, N=N, K=K, bias=dev(b), c_mode=4, C2=C2)  # raw to C, ELU copy to C2
    close(C, ref, 5e-5, "raw")
    close(C2, F.elu(ref), 5e-5, "activated copy")
    wg, bg = pack.pack_glu(W, b)
    G = torch.empty(M, N // 2, device=DEV)
    hip.gemm(dev(A), hip.pack_w_bf16x1(dev(wg)), G, M=M, N=N, K=K, bias=dev(bg), epilogue=hip.EPI_GLU)
    close(G, ref[:, : N // 2] * torch.sigmoid(ref[:, N // 2:]), 3e-5, "glu")
    # fused RMSNorm: the raw row is what gets rounded, the row scale multiplies the accumulator
    eps, nw = 1e-6, 1.0 + 0.3 * rnd(K, seed=54)
    Wf = (W * nw[None, :]).contiguous()
    rs = torch.rsqrt((A.double() ** 2).mean(-1, keepdim=True) + eps)
    refn = ((bf(A).double() @ bf(Wf).double().t()) * rs + b.double()).float()
    hip.gemm(dev(A), hip.pack_w_bf16x1(dev(Wf)), C, M=M, N=N, K=K, bias=dev(b), rms_eps=eps)
    close(C, refn, 5e-5, "rmsnorm+gemm")
    # causal conv through overlapping row windows (k = 3): the SEANet addressing
    B, T, ci, co, k = 2, 50, 64, 96, 3
    x, w3, b3 = rnd(B, T, ci, seed=11), rnd(co, ci, k, seed=12, scale=(ci * k) ** -0.5), rnd(co, seed=13)
    refc = F.conv1d(F.pad(bf(x).transpose(1, 2), (k - 1, 0)), bf(w3), b3).transpose(1, 2)
    buf = torch.zeros(B, k - 1 + T, ci)
    buf[:, k - 1:] = x
    wrow = w3.permute(0, 2, 1).reshape(co, k * ci).contiguous()
    out = torch.empty(B * T, co, device=DEV)
    hip.gemm(dev(buf), hip.pack_w_bf16x1(dev(wrow)), out, M=B * T, N=co, K=k * ci, lda=ci, bias=dev(b3), rows_per_seg=T, a_seg_stride=(k - 1 + T) * ci)
    close(out.view(B, T, co), refc, 5e-5, "row-window conv")
    with pytest.raises(hip.SoproHipError):
        hip.gemm(dev(A), Wp, C, M=M, N=N, K=K, a_split=True)


@pytest.mark.parametrize("B", [1, 16, 32, 37])
def test_skinny_bf16_weights_match_the_rounded_product(B, w):
    """AR-step contractions in bf16 mode (w_layout 2): bf16 weights in fragment order, activations rounded to bf16 as MFMA
    operands, fp32 accumulate - against torch on the rounded operands; RMSNorm statistics, bias, GELU, the K-split residual
    form and the GLU / ring-buffer tail stay fp32."""
    D = 384
    # RMSNorm -> projection (+ bias, GELU): the row scale comes from the raw fp32 row, the product from the rounded one
    X, nw, W, b = rnd(B, D, seed=20), 1 + 0.1 * rnd(D, seed=21), rnd(4 * D, D, seed=22, scale=D ** -0.5), rnd(4 * D, seed=23)
    Wf = (W * nw[None, :]).contiguous()
    rs = torch.rsqrt((X.double() ** 2).mean(-1, keepdim=True) + 1e-6)
    ref = ((bf(X).double() @ bf(Wf).double().t()) * rs + b.double()).float()
    Y = torch.full((B, 4 * D), float("nan"), device=DEV)
    hip.skinny(dev(X), hip.pack_skinny_w(dev(Wf), bf16=True), Y, B=B, N=4 * D, K=D, rms_norm=True, eps=1e-6, bias=dev(b), epilogue=hip.EPI_GELU)
    close(Y, F.gelu(ref), 5e-5, "bf16 ff1")
    # head (N = 2049: a ragged last tile)
    Wh, bh = rnd(2049, D, seed=24, scale=D ** -0.5), rnd(2049, seed=25)
    Yh = torch.full((B, 2049), float("nan"), device=DEV)
    hip.skinny(dev(X), hip.pack_skinny_w(dev(Wh), bf16=True), Yh, B=B, N=2049, K=D, rms_norm=True, eps=1e-6, bias=dev(bh))
    close(Yh, ((bf(X).double() @ bf(Wh).double().t()) * rs + bh.double()).float(), 5e-5, "bf16 head")
    # K = 1536 as four K-slices, slice 0 carrying bias + residual
    U, W2, b2, R = rnd(B, 4 * D, seed=26), rnd(D, 4 * D, seed=27, scale=(4 * D) ** -0.5), rnd(D, seed=28), rnd(B, D, seed=29)
    P = torch.full((4, B, D), float("nan"), device=DEV)
    hip.skinny(dev(U), hip.pack_skinny_w(dev(W2), bf16=True), P, B=B, N=D, K=4 * D, bias=dev(b2), epilogue=hip.EPI_RES, R=dev(R), ksplit=True,
               y_part_stride=B * D)
    close(P.sum(0), R + b2 + bf(U) @ bf(W2).t(), 1e-4, "bf16 ff2 slices")
    # GLU -> ring buffer -> taps -> + x, on an empty ring: only the newest tap contributes
    p, k, dil = "ar.blocks.1", 13, 2
    L = (k - 1) * dil + 1
    ring = torch.zeros(L, B, D, device=DEV)
    step = torch.zeros(1, dtype=torch.int32, device=DEV)
    gw = (w[p + ".glu.pro.weight"] * w[p + ".norm.weight"][None, :]).contiguous()
    Yg = torch.empty(B, D, device=DEV)
    hip.skinny(dev(X), hip.pack_skinny_w(dev(gw), glu=True, bf16=True), Yg, B=B, N=2 * D, K=D, rms_norm=True, eps=1e-6, bias=dev(w[p + ".glu.pro.bias"]),
               epilogue=hip.EPI_GLU_DW, ring=ring, dw_w=dev(pack.pack_dw(w[p + ".dw.dw.weight"])), dw_b=dev(w[p + ".dw.dw.bias"]), step=step,
               ring_len=L, ring_bcap=B, dil=dil, ksize=k)
    pre = ((bf(X).double() @ bf(gw).double().t()) * rs).float() + w[p + ".glu.pro.bias"]
    h = pre[:, :D] * torch.sigmoid(pre[:, D:])
    close(Yg, X + h * w[p + ".dw.dw.weight"][:, 0, -1] + w[p + ".dw.dw.bias"], 1e-4, "bf16 glu tail")
    close(ring[0], h, 5e-5, "ring row of this frame")


@pytest.mark.parametrize("S,np_", [(19, 0), (64, 3), (130, 3)])
def test_xattn_step_bf16_operands_match_the_rounded_operands(S, np_, w):
    """Round 4: the folded text operands K' / V' of the AR frame stored as bf16 (sopro_xattn_args.kv_format 1) - against the
    fp32-operand kernel on the bf16-ROUNDED operands (the same arithmetic on the same values: round-off class) and against
    TextXAttnBlock on the unrounded ones (src/sopro/nn/text.py:85-132; a bf16-class deviation)."""
    B, H, D = 5, 4, 384
    dh = D // H
    p = "ar.x_attns.3"
    S_cap = ((S + 63) // 64) * 64
    parts = [rnd(B, D, seed=800 + i) for i in range(np_ + 1)]
    x = sum(parts)
    ctx = rnd(B, S, D, seed=810)
    klens = [S, 1, max(1, S // 2), S, max(1, S - 3)]
    keep = torch.arange(S)[None, :] < torch.tensor(klens)[:, None]
    k, v = O.xattn_kv(ctx, w, p, H)
    ref = O.text_xattn(x[:, None], k, v, keep, w, p)[:, 0]
    Wq, Wo = w[p + ".q_proj.weight"], w[p + ".out_proj.weight"]
    Kp, Vp = torch.zeros(B, H, S_cap, D), torch.zeros(B, H, S_cap, D)
    for h in range(H):
        Kp[:, h, :S] = (k[:, h] @ Wq[h * dh:(h + 1) * dh]) * w[p + ".nq.weight"]
        Vp[:, h, :S] = v[:, h] @ Wo[:, h * dh:(h + 1) * dh].t()
    Pd = dev(torch.stack(parts))
    kl = dev(torch.tensor(klens, dtype=torch.int32))
    kw = dict(B=B, H=H, D=D, S_cap=S_cap, gate=float(torch.tanh(w[p + ".gate"])), scale=dh ** -0.5, eps=1e-6, Xp=Pd[1:] if np_ else None, np_=np_,
              xp_stride=B * D, y_part_stride=B * D)
    K16, V16 = torch.empty(B, H, S_cap, D, dtype=torch.bfloat16, device=DEV), torch.empty(B, H, S_cap, D, dtype=torch.bfloat16, device=DEV)
    hip.cvt_f32_bf16(dev(Kp), K16)
    hip.cvt_f32_bf16(dev(Vp), V16)
    torch.cuda.synchronize()
    assert torch.equal(K16.cpu(), Kp.to(torch.bfloat16)) and torch.equal(V16.cpu(), Vp.to(torch.bfloat16)), "sopro_cvt_f32_bf16 is not round-to-nearest-even"
    back = torch.empty(B, H, S_cap, D, device=DEV)
    hip.cvt_bf16_f32(K16, back)
    assert torch.equal(back.cpu(), bf(Kp))
    Y16 = torch.full((H, B, D), float("nan"), device=DEV)
    hip.xattn_step(Pd[0], Y16, None, K16, V16, kl, **kw)
    Y32 = torch.full((H, B, D), float("nan"), device=DEV)
    hip.xattn_step(Pd[0], Y32, None, dev(bf(Kp)), dev(bf(Vp)), kl, **kw)
    close(Y16.sum(0), Y32.sum(0), 2e-5, "bf16-stored operands vs the same values in fp32")
    close(Y16.sum(0), ref, 3e-2, "bf16 operands vs the unrounded block")
    assert float((Y16.sum(0).cpu() - ref).abs().max()) > 1e-6  # (the operands really were rounded)


@pytest.mark.parametrize("B,dil", [(5, 2), (32, 4)])
def test_skinny_glu_bf16_ring_buffer_over_frames(B, dil, w):
    """Round 4: the AR frame's ring buffers stored as bf16 (sopro_skinny_args.ring_format 1, bf16 mode only): h is rounded once
    when it is written and the twelve older taps are widened when they are read - over 30 frames against a torch model of
    exactly that (bf16 weights / operands as in w_layout 2, the newest tap in fp32, older taps bf16(h))."""
    D, k = 384, 13
    p = "ar.blocks.2"
    L = (k - 1) * dil + 1
    ring = torch.zeros(L, B, D, dtype=torch.bfloat16, device=DEV)
    step = torch.zeros(1, dtype=torch.int32, device=DEV)
    gwf = (w[p + ".glu.pro.weight"] * w[p + ".norm.weight"][None, :]).contiguous()
    gw, gb = hip.pack_skinny_w(dev(gwf), glu=True, bf16=True), dev(w[p + ".glu.pro.bias"])
    dww, dwb = dev(pack.pack_dw(w[p + ".dw.dw.weight"])), dev(w[p + ".dw.dw.bias"])
    wt = w[p + ".dw.dw.weight"].squeeze(1)  # [D, k], oldest tap first
    Y = torch.empty(B, D, device=DEV)
    hist = torch.zeros(B, L, D)  # bf16-rounded h of earlier frames, oldest first
    for t in range(30):
        x = rnd(B, D, seed=100 + t)
        rs = torch.rsqrt((x.double() ** 2).mean(-1, keepdim=True) + 1e-6)
        pre = ((bf(x).double() @ bf(gwf).double().t()) * rs).float() + w[p + ".glu.pro.bias"]
        h = pre[:, :D] * torch.sigmoid(pre[:, D:])
        hist = torch.cat([hist[:, 1:], h.unsqueeze(1)], dim=1)
        taps = hist[:, torch.arange(0, k * dil, dil)].clone()  # [B, k, D]; the newest (last) is this frame's fp32 h
        y = (taps.transpose(1, 2) * wt).sum(-1) + w[p + ".dw.dw.bias"]
        hist[:, -1] = bf(h)  # what later frames read back
        step.fill_(t)
        hip.skinny(dev(x), gw, Y, B=B, N=2 * D, K=D, rms_norm=True, eps=1e-6, bias=gb, epilogue=hip.EPI_GLU_DW, ring=ring,
                   dw_w=dww, dw_b=dwb, step=step, ring_len=L, ring_bcap=B, dil=dil, ksize=k)
        close(Y, x + y, 5e-3, f"glu + bf16 ring, frame {t}")  # (h itself carries the bf16 operands' round-off: a torch-vs-MFMA order effect at the rounding boundaries)
    torch.cuda.synchronize()
    assert ring.dtype == torch.bfloat16 and float(ring.float().abs().max()) > 0
    with pytest.raises(hip.SoproHipError):  # a bf16 ring needs the bf16 weights of the mode
        hip.skinny(dev(x), hip.pack_skinny_w(dev(gwf), glu=True), Y, B=B, N=2 * D, K=D, rms_norm=True, eps=1e-6, bias=gb, epilogue=hip.EPI_GLU_DW,
                   ring=ring, dw_w=dww, dw_b=dwb, step=step, ring_len=L, ring_bcap=B, dil=dil, ksize=k)


# ----------------------------------------------------------------------- round 4: bf16 ROWS in memory (the decoder's activation flow)
def b16(t):
    return t.to(torch.bfloat16)


def test_gemm_bf16x1_bf16_rows_in_and_out():
    """sopro_gemm_bf16x1 with a_format 2 (A = bf16 rows, staged by plain copies) and c_mode 6 / 7 / 8 (bf16 rows out, raw /
    activated / both), EPI_RES with a bf16 skip operand, row windows (the SEANet addressing), fp32 rows in -> bf16 rows out
    (the first convolution), a ragged M / odd tile counts, and split-K on a few rows (streaming chunks).  Reference: torch on
    the bf16 values in fp32, rounded to bf16 where the kernel rounds."""
    M, N, K = 333, 192, 256
    A, W, b, R = rnd(M, K, seed=41), rnd(N, K, seed=42, scale=K ** -0.5), rnd(N, seed=43), rnd(M, N, seed=44)
    Wp = hip.pack_w_bf16x1(dev(W))
    A16, R16 = dev(b16(A)), dev(b16(R))
    ref = bf(A) @ bf(W).t() + b  # fp32 accumulation of exact products
    # bf16 in, fp32 out: the same numbers as the fp32-row kernel gives for the rounded values
    C = torch.full((M, N), float("nan"), device=DEV)
    hip.gemm(A16, Wp, C, M=M, N=N, K=K, bias=dev(b))
    C0 = torch.empty(M, N, device=DEV)
    hip.gemm(dev(bf(A)), Wp, C0, M=M, N=N, K=K, bias=dev(b))
    assert torch.equal(C, C0), "bf16 rows must give what the fp32 rows of the same values give (same MFMA operands, same order)"
    close(C, ref, 5e-5, "bf16 A, fp32 C")

    def near_bf16(got16, want32, what):  # equal up to one rounding step at a boundary
        g, w_ = got16.float().cpu(), want32.float()
        tol = w_.abs() * 2.0 ** -7 + 1e-30
        bad = (g - bf(w_)).abs() > tol
        assert not bool(bad.any()), f"{what}: {int(bad.sum())} elements off by more than one bf16 step"
        assert float((g != bf(w_)).float().mean()) < 0.01, f"{what}: too many elements differ from the rounded reference"

    C6 = torch.full((M, N), float("nan"), dtype=torch.bfloat16, device=DEV)
    hip.gemm(A16, Wp, C6, M=M, N=N, K=K, bias=dev(b), c_mode=6)
    near_bf16(C6, ref, "c_mode 6 (raw bf16)")
    C7 = torch.full((M, N), float("nan"), dtype=torch.bfloat16, device=DEV)
    hip.gemm(A16, Wp, C7, M=M, N=N, K=K, bias=dev(b), c_mode=7)
    near_bf16(C7, F.elu(ref), "c_mode 7 (activated bf16)")
    C8, C8a = torch.full((M, N), float("nan"), dtype=torch.bfloat16, device=DEV), torch.full((M, N), float("nan"), dtype=torch.bfloat16, device=DEV)
    hip.gemm(A16, Wp, C8, M=M, N=N, K=K, bias=dev(b), c_mode=8, C2=C8a)
    assert torch.equal(C8, C6) and torch.equal(C8a, C7)
    # residual form: skip operand in bf16, sum in fp32, ELU, rounded once
    Cr = torch.full((M, N), float("nan"), dtype=torch.bfloat16, device=DEV)
    hip.gemm(A16, Wp, Cr, M=M, N=N, K=K, bias=dev(b), epilogue=hip.EPI_RES, R=R16, c_mode=7)
    near_bf16(Cr, F.elu(bf(R) + ref), "EPI_RES + c_mode 7")
    # fp32 rows in (the transformer stream), activated bf16 rows out: the first convolution of the decoder
    Cf = torch.full((M, N), float("nan"), dtype=torch.bfloat16, device=DEV)
    hip.gemm(dev(A), Wp, Cf, M=M, N=N, K=K, bias=dev(b), c_mode=7)
    near_bf16(Cf, F.elu(ref), "fp32 A, c_mode 7")
    # causal conv through overlapping bf16 row windows with padded segments on both sides
    B, T, ci, co, k = 2, 150, 64, 96, 3
    x, w3, b3 = rnd(B, T, ci, seed=51), rnd(co, ci, k, seed=52, scale=(ci * k) ** -0.5), rnd(co, seed=53)
    refc = F.conv1d(F.pad(bf(x).transpose(1, 2), (k - 1, 0)), bf(w3), b3).transpose(1, 2)
    buf = torch.zeros(B, k - 1 + T + 3, ci)
    buf[:, k - 1:k - 1 + T] = x
    wrow = w3.permute(0, 2, 1).reshape(co, k * ci).contiguous()
    out = torch.full((B, 2 + T + 1, co), float("nan"), dtype=torch.bfloat16, device=DEV)
    hip.gemm(dev(b16(buf)), hip.pack_w_bf16x1(dev(wrow)), out, M=B * T, N=co, K=k * ci, lda=ci, bias=dev(b3), rows_per_seg=T, a_seg_stride=(k - 1 + T + 3) * ci,
             c_off=2 * co, c_seg_stride=(2 + T + 1) * co, ldc=co, c_mode=6)
    near_bf16(out[:, 2:2 + T], refc, "bf16 row-window conv")
    assert bool(torch.isnan(out[:, :2].float()).all()) and bool(torch.isnan(out[:, 2 + T:].float()).all())  # nothing outside its rows
    # a few rows with a long K: the split-K path (streaming chunks of the decoder)
    M2, N2, K2 = 12, 512, 2048
    A2, W2, b2 = rnd(M2, K2, seed=61), rnd(N2, K2, seed=62, scale=K2 ** -0.5), rnd(N2, seed=63)
    C2o, C2a = torch.full((M2, N2), float("nan"), dtype=torch.bfloat16, device=DEV), torch.full((M2, N2), float("nan"), dtype=torch.bfloat16, device=DEV)
    hip.gemm(dev(b16(A2)), hip.pack_w_bf16x1(dev(W2)), C2o, M=M2, N=N2, K=K2, bias=dev(b2), c_mode=8, C2=C2a)
    r2 = bf(A2) @ bf(W2).t() + b2
    near_bf16(C2o, r2, "split-K raw")
    near_bf16(C2a, F.elu(r2), "split-K activated")
    with pytest.raises(hip.SoproHipError):  # bf16 rows take no activation prologue
        hip.gemm(A16, Wp, C6, M=M, N=N, K=K, prologue=hip.PRO_ELU, c_mode=6)
    with pytest.raises(hip.SoproHipError):  # ... and belong to the one-pass path
        hip.gemm(A16, hip.pack_w_bf16x3(dev(W)), C, M=M, N=N, K=K)


def _tail_ref(h16, w1, b1, w2, b2, wf, bf_):
    """torch model of the bf16-row tail: h bf16 -> ELU -> bf16 -> conv -> +b -> ELU -> bf16 -> conv -> + skip(h) + b -> ELU (fp32) -> last conv."""
    x = h16.float().transpose(1, 2)
    y = O.causal_conv1d(bf(F.elu(x)), bf(w1), b1)
    y = O.causal_conv1d(bf(F.elu(y)), bf(w2), b2)
    return O.causal_conv1d(F.elu(x + y), wf, torch.tensor([bf_]))[:, 0]


@pytest.mark.parametrize("B,T", [(2, 300), (1, 13), (3, 4100)])
def test_seanet_tail_on_bf16_rows(B, T):
    h = rnd(B, T, 64, seed=730)
    w1, b1 = rnd(32, 64, 3, seed=731, scale=0.07), rnd(32, seed=732, scale=0.1)
    w2, b2 = rnd(64, 32, 1, seed=733, scale=0.17), rnd(64, seed=734, scale=0.1)
    wf, bf_ = rnd(1, 64, 3, seed=735, scale=0.07), 0.03
    hb = torch.zeros(B, 2 + T, 64, dtype=torch.bfloat16)
    hb[:, 2:] = b16(h)
    ref = _tail_ref(hb[:, 2:], w1, b1, w2, b2, wf, bf_)
    wav = torch.full((B, T), float("nan"), device=DEV)
    hip.seanet_tail_bf16(dev(hb), dev(pack.pack_conv1d(w1)), dev(b1), dev(pack.pack_conv1d(w2)), dev(b2), dev(wf[0].t()), bf_, wav,
                         B=B, T=T, h_seg_stride=(2 + T) * 64, wav_seg_stride=T)
    # (the intermediate's bf16 rounding can fall on the other side of a boundary than torch's: 2^-8 of one of 32 operands)
    close(wav, ref, 3e-3 * float(ref.abs().max() + 1.0), "tail on bf16 rows")
    # and it is a bf16-class approximation of the fp32 layers
    x = h.transpose(1, 2)
    y = O.causal_conv1d(F.elu(O.causal_conv1d(F.elu(x), w1, b1)), w2, b2)
    full = O.causal_conv1d(F.elu(x + y), wf, torch.tensor([bf_]))[:, 0]
    close(wav, full, 3e-2 * float(full.abs().max() + 1.0), "tail on bf16 rows vs fp32 layers")


def test_seanet_res128_on_bf16_rows():
    B, T = 3, 333
    h = rnd(B, T, 128, seed=720)
    w1, b1 = rnd(64, 128, 3, seed=721, scale=0.05), rnd(64, seed=722, scale=0.1)
    w2, b2 = rnd(128, 64, 1, seed=723, scale=0.12), rnd(128, seed=724, scale=0.1)
    hb = torch.zeros(B, 2 + T + 5, 128, dtype=torch.bfloat16)
    hb[:, 2:2 + T] = b16(h)
    x = hb[:, 2:2 + T].float().transpose(1, 2)
    y = O.causal_conv1d(bf(F.elu(x)), bf(w1), b1)
    y = O.causal_conv1d(bf(F.elu(y)), bf(w2), b2)
    ref = F.elu(x + y).transpose(1, 2)
    args = [dev(hb)] + [dev(t) for t in (pack.pack_conv1d(w1), b1, pack.pack_conv1d(w2), b2)]
    lib, outs = hip.load(), []
    try:
        for tiles in (0, 1, 4):
            lib.sopro_seanet_res_set_tiles(tiles)
            out = torch.full((B, 2 + T + 5, 128), float("nan"), dtype=torch.bfloat16, device=DEV)
            out[:, :2] = 0.0
            hip.seanet_res128_bf16(*args, out, B=B, T=T, h_seg_stride=(2 + T + 5) * 128, out_seg_stride=(2 + T + 5) * 128)
            torch.cuda.synchronize()
            outs.append(out.cpu())
    finally:
        lib.sopro_seanet_res_set_tiles(0)
    got = outs[0][:, 2:2 + T].float()
    assert float((got - ref).abs().max()) <= 2.0 ** -6 * float(ref.abs().max()) + 1e-3  # a bf16 step of the result + the intermediate's boundary cases
    assert float((got - ref).abs().mean()) < 2.0 ** -9 * float(ref.abs().mean() + 1e-3) * 4
    assert bool(torch.isnan(outs[0][:, 2 + T:].float()).all()) and float(outs[0][:, :2].float().abs().max()) == 0.0
    for o in outs[1:]:
        assert torch.equal(o[:, 2:2 + T], outs[0][:, 2:2 + T])


def test_seanet_up128_on_bf16_rows_equals_the_tile_kernel():
    """The weight-stationary transposed convolution on bf16 rows: bit for bit what sopro_gemm_bf16x1 (a_format 2, c_mode 6) gives
    on the same operands, for any number of tiles per workgroup; and torch's conv_transpose1d on the rounded operands."""
    B, T, ci, co, r = 3, 333, 128, 64, 4
    x = rnd(B, T, ci, seed=760)
    wt, bt = rnd(ci, co, 2 * r, seed=761, scale=0.06), rnd(co, seed=762, scale=0.1)
    W, bias = pack.pack_convtr1d(wt, bt, r)
    xs, os_ = (1 + T + 7) * ci, (2 + T * r + 6) * co
    xb = torch.zeros(B, 1 + T + 7, ci, dtype=torch.bfloat16)
    xb[:, 1:1 + T] = b16(x)
    ref = F.conv_transpose1d(xb[:, 1:1 + T].float().transpose(1, 2), bf(wt), bt, stride=r)[..., :T * r].transpose(1, 2)
    xd, Wd, bd = dev(xb), dev(W), dev(bias)
    lib, outs = hip.load(), []
    try:
        for tiles in (0, 1, 2, 5):
            lib.sopro_seanet_up_set_tiles(tiles)
            out = torch.full((B, 2 + T * r + 6, co), float("nan"), dtype=torch.bfloat16, device=DEV)
            hip.seanet_up128_bf16(xd, Wd, bd, out, B=B, T=T, x_seg_stride=xs, out_seg_stride=os_, out_off=2 * co)
            torch.cuda.synchronize()
            outs.append(out.cpu())
    finally:
        lib.sopro_seanet_up_set_tiles(0)
    got = outs[0][:, 2:2 + T * r]
    assert float((got.float() - ref).abs().max()) <= 2.0 ** -7 * float(ref.abs().max()) + 1e-4
    assert bool(torch.isnan(outs[0][:, :2].float()).all()) and bool(torch.isnan(outs[0][:, 2 + T * r:].float()).all())
    for o in outs[1:]:
        assert torch.equal(o[:, 2:2 + T * r], got)
    out2 = torch.full((B, 2 + T * r + 6, co), float("nan"), dtype=torch.bfloat16, device=DEV)
    hip.gemm(xd, hip.pack_w_bf16x1(Wd), out2, M=B * T, N=r * co, K=2 * ci, lda=ci, bias=bd, rows_per_seg=T, a_seg_stride=xs, c_off=2 * co, c_seg_stride=os_,
             ldc=r * co, c_mode=6)
    torch.cuda.synchronize()
    assert torch.equal(out2.cpu()[:, 2:2 + T * r], got)


def test_mimi_decode_bf16_rows_against_the_fp32_rows_form(cfg, mimi_np, mc):
    """The whole decoder in bf16 mode with bf16 activations in memory against the same mode with fp32 activations (SOPRO_MIMI_BF16=0,
    the round-3 form: operands rounded in flight) and against the fp32 engine: the new storage must cost little on top of what the
    mode's one-pass products already cost.  Batch and streaming chunk shapes."""
    import os
    import subprocess
    import sys

    g = golden("full200")
    from sopro_amd.codec import MimiCodec
    from sopro_amd.config import MimiDecoderConfig

    codes = torch.from_numpy(g["tokens"].astype(np.int64))[:60]
    c16 = MimiCodec(mimi_np, MimiDecoderConfig(num_quantizers=32), DEV, precision="bf16")
    c32 = MimiCodec(mimi_np, MimiDecoderConfig(num_quantizers=32), DEV, precision="f32")
    w16 = c16.decode_full(codes).float().cpu().reshape(-1)
    w32 = c32.decode_full(codes).float().cpu().reshape(-1)
    wb = c16.decode_batch(torch.stack([codes, codes.flip(0), codes]))
    # (few-row launches take split-K by row count: another summation order in front of the roundings, so a batch row is the lone
    # run's waveform up to the mode's own rounding noise - measured 7e-3 of peak - not bit for bit)
    assert float((wb[0].cpu() - w16).abs().max()) <= 2e-2 * float(w32.abs().max())
    snr = 10.0 * float(torch.log10((w32 ** 2).mean() / ((w16 - w32) ** 2).mean()))
    # the fp32-rows form of the same mode, in a child process (the switch is read once per process)
    code = ("import sys, numpy as np, torch; sys.path.insert(0, %r); sys.path.insert(0, %r + '/tests')\n"
            "from conftest import golden, SEED\nfrom sopro_amd.codec import MimiCodec\nfrom sopro_amd.config import MimiDecoderConfig\n"
            "from sopro_amd.weights import synth_mimi_weights\nmc = MimiDecoderConfig(num_quantizers=32)\n"
            "c = MimiCodec(synth_mimi_weights(MimiDecoderConfig(), SEED), mc, 'cuda:0', precision='bf16')\n"
            "codes = torch.from_numpy(golden('full200')['tokens'].astype(np.int64))[:60]\n"
            "np.save(sys.argv[1], c.decode_full(codes).float().cpu().numpy())\n") % (os.path.dirname(os.path.dirname(os.path.abspath(__file__))),
                                                                                   os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import tempfile
    path = os.path.join(tempfile.mkdtemp(), "w.npy")
    subprocess.check_call([sys.executable, "-c", code, path], env=dict(os.environ, SOPRO_MIMI_BF16="0"), timeout=600)
    wf32rows = torch.from_numpy(np.load(path)).reshape(-1)
    snr_old = 10.0 * float(torch.log10((w32 ** 2).mean() / ((wf32rows - w32) ** 2).mean()))
    print(f"\nbf16 mode decoder vs fp32 engine (60 frames): bf16 rows in memory SNR {snr:.1f} dB; fp32 rows (round-3 form) {snr_old:.1f} dB")
    assert snr > 38.0 and snr > snr_old - 3.0


@pytest.fixture(scope="module")
def tts_bf16(cfg, sopro_np_noeos, mimi_np):
    from sopro_amd import SoproTTS
    return SoproTTS.from_weights(cfg, sopro_np_noeos, mimi_np, FakeTok(), device="cuda:0", precision="bf16")


def test_bf16_mode_quality_against_the_fp32_reference(tts_bf16, cfg, mc, w_noeos):
    # (prep below runs on the bf16 engine: conditioning is fp32-class in both modes)
    """On the reference's 200-frame fixture (S = 64, Tr = 150): refined-token agreement with the fp32 reference given its
    codebook-0 tokens, the oracle's own logit gap where the bf16 arg-max differs, and the waveform error of the bf16
    decoder on the reference's tokens.  Numbers go to the test log (DESIGN.md quotes them); the floors only catch a broken path."""
    g = golden("full200")
    tts = tts_bf16
    want = torch.from_numpy(g["tokens"].astype(np.int64))
    T = int(want.shape[0])
    ref = tts.prepare_reference(ref_tokens_tq=torch.from_numpy(g["ref_tq"]))
    prep = tts.model.prepare_conditioning(torch.from_numpy(g["ids"]), ref, max_frames=int(g["max_frames"]), style_strength=1.0)
    got = tts.model.nar_refine(prep["cond_ar"][:, :T], want[:, 0].unsqueeze(0))[0].cpu()
    agree = float((got[:, 1:] == want[:, 1:]).float().mean())
    oref = O.prepare_reference(torch.from_numpy(g["ref_tq"]), w_noeos, cfg)
    oprep = O.prepare_conditioning(torch.from_numpy(g["ids"]), oref, w_noeos, cfg, max_frames=int(g["max_frames"]), style_strength=1.0)
    cond_err = float((prep["cond_ar"].cpu() - oprep["cond_ar"]).abs().max())
    n_off, gap = O.nar_audit(oprep["cond_ar"][:, :T], got.unsqueeze(0), w_noeos, cfg)
    wav = tts.codec.decode_full(want).cpu().reshape(-1)
    ref_wav = torch.from_numpy(g["wav"])
    werr = float((wav - ref_wav).abs().max()) / float(ref_wav.abs().max())
    snr = 10.0 * float(torch.log10((ref_wav ** 2).mean() / ((wav - ref_wav) ** 2).mean()))
    # the AR frame streams bf16 weights in this mode: teacher-forced on the reference's own codebook-0 tokens, how often is the
    # bf16 frame's arg-max the reference's next token, and how far are its logits from the fp32 oracle's?
    from sopro_amd.model import _ARRun

    m = tts.model
    run = _ARRun(m, prep["cond_ar"], prep["txt_seq"], None, top_p=0.0, temperature=1.0, anti_loop=False, min_gen_frames=10 ** 6)
    ost = O.ar_init_state(1, oprep["txt_seq"], oprep["text_mask"], w_noeos, cfg)
    E = w_noeos["cb_embed.emb.weight"]
    hits, worst_lg, hist = 0, 0.0, []
    for t in range(T):
        prev = E[int(cfg.num_codebooks) * int(cfg.codebook_size)] if t == 0 else E[int(want[t - 1, 0])]
        x = oprep["cond_ar"][:, t, :] + prev.unsqueeze(0)
        olog = O.ar_step(x, ost, w_noeos, cfg)[0]
        with torch.cuda.stream(m.stream):
            run.plan.x[0].copy_(x.to(m.device))
        run.advance(1)
        m.stream.synchronize()
        glog = run.plan.logits[0].cpu()
        worst_lg = max(worst_lg, float((glog - olog).abs().max()))
        hits += int(int(O.penalised_logits(glog, hist, 1.0, 1.1).argmax()) == int(want[t, 0]))
        hist.append(int(want[t, 0]))
    run.done = True
    ar_agree = hits / T
    print(f"\nbf16 mode vs fp32 reference (full200): cond_ar max err {cond_err:.2e}; AR teacher-forced: next-token agreement {ar_agree:.4f}, "
          f"max |dlogit| {worst_lg:.3f}; refined-token agreement {agree:.4f} ({n_off} of {T * 31} off the fp32 arg-max, worst fp32 logit "
          f"gap {gap:.3f}); waveform max err {werr:.3e} of peak, SNR {snr:.1f} dB")
    # The mode's stated gate: teacher-forced AR next-token agreement >= 98.5 % (measured 99.0 %: 198 of 200
    # frames), AR logits within 3e-2 of the fp32 oracle's (measured 0.023; SURVEY 8c estimated 2e-2 before the bf16 weight stream
    # existed), waveform SNR >= 40 dB against the fp32 reference on its own tokens (measured 41.8 dB with bf16 rows in memory,
    # 42.6 dB with fp32 rows).  Refined tokens are reported, with a floor: one early flip changes the later stages' inputs, so
    # equality with the reference's 6200 tokens (0.84) says less than the fp32 logit gap at the flips (worst 0.012).
    assert ar_agree >= 0.985 and worst_lg <= 3e-2 and snr >= 40.0 and werr < 0.02
    assert agree > 0.80 and gap < 0.05
