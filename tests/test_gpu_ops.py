"""Kernel-level parity (MI355X): every entry point of libsopro_hip.so against the CPU oracle / plain
torch fp32 on the same seeded inputs, through the C ABI (sopro_amd/hip.py is marshalling only).
Tolerances are fp32 round-off class: the kernels accumulate in fp32 (exact-f32 MFMA), only the
summation order differs from the CPU."""
import ctypes as C
import math

import numpy as np
import pytest
import torch
import torch.nn.functional as F

from conftest import golden
from oracle import sopro_oracle as O
from sopro_amd import hip, pack

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def rnd(*shape, seed=0, scale=1.0):
    g = torch.Generator().manual_seed(seed)
    return torch.randn(*shape, generator=g) * scale


def dev(t):
    return t.to(DEV).contiguous()


def _t(a):
    return torch.from_numpy(np.asarray(a))


def close(a, b, atol, what=""):
    a, b = a.detach().float().cpu(), b.detach().float().cpu()
    err = float((a - b).abs().max())
    assert err <= atol, f"{what}: max abs err {err:.3e} > {atol:.1e}"


# --------------------------------------------------------------------------------------------- GEMM
@pytest.mark.parametrize("M,N,K", [(300, 200, 96), (129, 64, 64), (1000, 32, 192), (70, 1, 192), (5, 384, 384), (257, 2048, 256)])
def test_gemm_plain_bias(M, N, K):
    A, W, b = rnd(M, K, seed=1), rnd(N, K, seed=2, scale=K ** -0.5), rnd(N, seed=3)
    C = torch.full((M, N), float("nan"), device=DEV)
    hip.gemm(dev(A), dev(W), C, M=M, N=N, K=K, bias=dev(b))
    torch.cuda.synchronize()
    close(C, A @ W.t() + b, 2e-5 * math.sqrt(K), f"gemm {M}x{N}x{K}")


def test_gemm_epilogues_and_prologues():
    M, N, K = 200, 256, 128
    A, W, b = rnd(M, K, seed=4), rnd(N, K, seed=5, scale=K ** -0.5), rnd(N, seed=6)
    R, sc, pv = rnd(M, N, seed=7), rnd(N, seed=8), rnd(K, seed=9)
    ref = A @ W.t() + b
    C = torch.empty(M, N, device=DEV)
    hip.gemm(dev(A), dev(W), C, M=M, N=N, K=K, bias=dev(b), epilogue=hip.EPI_GELU)
    close(C, F.gelu(ref), 3e-5, "gelu")
    hip.gemm(dev(A), dev(W), C, M=M, N=N, K=K, bias=dev(b), epilogue=hip.EPI_TANH)
    close(C, torch.tanh(ref), 3e-5, "tanh")
    hip.gemm(dev(A), dev(W), C, M=M, N=N, K=K, bias=dev(b), epilogue=hip.EPI_RES, R=dev(R), scale=dev(sc))
    close(C, R + sc * ref, 5e-5, "res+scale")
    Rd = dev(R)
    hip.gemm(dev(A), dev(W), Rd, M=M, N=N, K=K, bias=dev(b), epilogue=hip.EPI_RES, R=Rd)  # in place
    close(Rd, R + ref, 5e-5, "res in place")
    hip.gemm(dev(A), dev(W), C, M=M, N=N, K=K, bias=dev(b), prologue=hip.PRO_ELU)
    close(C, F.elu(A) @ W.t() + b, 5e-5, "elu prologue")
    hip.gemm(dev(A), dev(W), C, M=M, N=N, K=K, bias=dev(b), prologue=hip.PRO_ADDVEC, pro_vec=dev(pv))
    close(C, (A + pv) @ W.t() + b, 5e-5, "addvec prologue")
    # GLU on packed value/gate blocks (reference: src/sopro/nn/blocks.py:16-23)
    wp, bp = pack.pack_glu(W, b)
    G = torch.empty(M, N // 2, device=DEV)
    hip.gemm(dev(A), dev(wp), G, M=M, N=N, K=K, bias=dev(bp), epilogue=hip.EPI_GLU)
    close(G, ref[:, : N // 2] * torch.sigmoid(ref[:, N // 2:]), 3e-5, "glu")


def test_gemm_causal_conv_and_convtranspose_through_row_windows():
    """Segmented, overlapping-row addressing == MimiConv1d / MimiConvTranspose1d (HF:modeling_mimi.py:210-405)."""
    B, T, ci, co, k = 3, 37, 64, 96, 7
    x = rnd(B, T, ci, seed=10)
    w, b = rnd(co, ci, k, seed=11, scale=(ci * k) ** -0.5), rnd(co, seed=12)
    ref = F.conv1d(F.pad(F.elu(x).transpose(1, 2), (k - 1, 0)), w, b).transpose(1, 2)
    buf = torch.zeros(B, k - 1 + T, ci)
    buf[:, k - 1:] = x
    out = torch.zeros(B, 2 + T, co, device=DEV)
    hip.gemm(dev(buf), dev(pack.pack_conv1d(w)), out, M=B * T, N=co, K=k * ci, lda=ci, bias=dev(b), prologue=hip.PRO_ELU,
             rows_per_seg=T, a_seg_stride=(k - 1 + T) * ci, c_off=2 * co, c_seg_stride=(2 + T) * co, ldc=co)
    close(out[:, 2:], ref, 5e-5, "causal conv1d")
    assert float(out[:, :2].abs().max()) == 0.0  # padding rows untouched
    # dense destination / dense residual with a segmented source (seg stride 0 == rows_per_seg * ld)
    dense = torch.empty(B * T, co, device=DEV)
    res = rnd(B * T, co, seed=15)
    hip.gemm(dev(buf), dev(pack.pack_conv1d(w)), dense, M=B * T, N=co, K=k * ci, lda=ci, bias=dev(b), prologue=hip.PRO_ELU,
             rows_per_seg=T, a_seg_stride=(k - 1 + T) * ci, epilogue=hip.EPI_RES, R=dev(res))
    close(dense.view(B, T, co), res.view(B, T, co) + ref, 5e-5, "causal conv1d, dense C and R")
    for s in (4, 5, 8):
        wt, bt = rnd(ci, co, 2 * s, seed=13 + s, scale=(2 * ci) ** -0.5), rnd(co, seed=14)
        y = F.conv_transpose1d(x.transpose(1, 2), wt, bt, stride=s)
        ref = y[..., : y.shape[-1] - s].transpose(1, 2)  # [B, T*s, co]
        wp, bp = pack.pack_convtr1d(wt, bt, s)
        buf = torch.zeros(B, 1 + T, ci)
        buf[:, 1:] = x
        out = torch.zeros(B, 2 + T * s, co, device=DEV)
        hip.gemm(dev(buf), dev(wp), out, M=B * T, N=s * co, K=2 * ci, lda=ci, bias=dev(bp), rows_per_seg=T,
                 a_seg_stride=(1 + T) * ci, c_off=2 * co, c_seg_stride=(2 + T * s) * co, ldc=s * co)
        close(out[:, 2:], ref, 5e-5, f"conv transpose stride {s}")



# ----------------------------------------------------------------------- split-bf16 GEMM (Mimi decoder path)
def _split_bound(A, W, b=None):
    """|err| <= 2^-15 * (|A| |W|^T): two bf16 halves keep 16 mantissa bits per operand, products accumulate in fp32."""
    mag = A.double().abs() @ W.double().abs().t()
    return mag * 2.0 ** -15 + 1e-6


@pytest.mark.parametrize("M,N,K", [(300, 200, 96), (129, 64, 64), (1000, 96, 192), (70, 33, 192), (5, 384, 384), (257, 2048, 256),
                                   (640, 640, 512), (256, 128, 36), (12, 1536, 512), (48, 512, 768), (7, 100, 64)])
def test_gemm_bf16x3_plain_bias(M, N, K):
    A, W, b = rnd(M, K, seed=1), rnd(N, K, seed=2, scale=K ** -0.5), rnd(N, seed=3)
    C = torch.full((M, N), float("nan"), device=DEV)
    Wp = hip.pack_w_bf16x3(dev(W))
    hip.gemm(dev(A), Wp, C, M=M, N=N, K=K, bias=dev(b))
    torch.cuda.synchronize()
    ref = A.double() @ W.double().t() + b.double()
    err = (C.cpu().double() - ref).abs()
    assert bool((err <= _split_bound(A, W)).all()), f"{M}x{N}x{K}: worst {float((err / _split_bound(A, W)).max()):.2f} of the bound"
    with pytest.raises(hip.SoproHipError):
        hip.gemm(dev(A), Wp, C, M=M, N=N + 1, K=K)


def test_gemm_bf16x3_epilogues_prologue_and_row_windows():
    M, N, K = 200, 256, 128
    A, W, b = rnd(M, K, seed=4), rnd(N, K, seed=5, scale=K ** -0.5), rnd(N, seed=6)
    R, sc = rnd(M, N, seed=7), rnd(N, seed=8)
    Wp = hip.pack_w_bf16x3(dev(W))
    ref = A @ W.t() + b
    C = torch.empty(M, N, device=DEV)
    hip.gemm(dev(A), Wp, C, M=M, N=N, K=K, bias=dev(b), epilogue=hip.EPI_GELU)
    close(C, F.gelu(ref), 2e-4, "gelu")
    hip.gemm(dev(A), Wp, C, M=M, N=N, K=K, bias=dev(b), epilogue=hip.EPI_RES, R=dev(R), scale=dev(sc))
    close(C, R + sc * ref, 3e-4, "res+scale")
    Rd = dev(R)
    hip.gemm(dev(A), Wp, Rd, M=M, N=N, K=K, bias=dev(b), epilogue=hip.EPI_RES, R=Rd)  # in place
    close(Rd, R + ref, 2e-4, "res in place")
    hip.gemm(dev(A), Wp, C, M=M, N=N, K=K, bias=dev(b), prologue=hip.PRO_ELU)
    close(C, F.elu(A) @ W.t() + b, 2e-4, "elu prologue")
    # causal conv / transposed conv through segmented overlapping rows, as the Mimi decoder issues them
    B, T, ci, co, k = 3, 37, 64, 96, 7
    x = rnd(B, T, ci, seed=10)
    w, bb = rnd(co, ci, k, seed=11, scale=(ci * k) ** -0.5), rnd(co, seed=12)
    ref = F.conv1d(F.pad(F.elu(x).transpose(1, 2), (k - 1, 0)), w, bb).transpose(1, 2)
    buf = torch.zeros(B, k - 1 + T, ci)
    buf[:, k - 1:] = x
    out = torch.zeros(B, 2 + T, co, device=DEV)
    hip.gemm(dev(buf), hip.pack_w_bf16x3(dev(pack.pack_conv1d(w))), out, M=B * T, N=co, K=k * ci, lda=ci, bias=dev(bb),
             prologue=hip.PRO_ELU, rows_per_seg=T, a_seg_stride=(k - 1 + T) * ci, c_off=2 * co, c_seg_stride=(2 + T) * co, ldc=co)
    close(out[:, 2:], ref, 2e-4, "causal conv1d")
    assert float(out[:, :2].abs().max()) == 0.0
    s_ = 5
    wt, bt = rnd(ci, co, 2 * s_, seed=18, scale=(2 * ci) ** -0.5), rnd(co, seed=14)
    y = F.conv_transpose1d(x.transpose(1, 2), wt, bt, stride=s_)
    ref = y[..., : y.shape[-1] - s_].transpose(1, 2)
    wp, bp = pack.pack_convtr1d(wt, bt, s_)
    buf = torch.zeros(B, 1 + T, ci)
    buf[:, 1:] = x
    out = torch.zeros(B, 2 + T * s_, co, device=DEV)
    hip.gemm(dev(buf), hip.pack_w_bf16x3(dev(wp)), out, M=B * T, N=s_ * co, K=2 * ci, lda=ci, bias=dev(bp), rows_per_seg=T,
             a_seg_stride=(1 + T) * ci, c_off=2 * co, c_seg_stride=(2 + T * s_) * co, ldc=s_ * co)
    close(out[:, 2:], ref, 2e-4, "conv transpose stride 5")


@pytest.mark.parametrize("K", [32, 96, 160, 16])
def test_gemm_split_odd_k_steps_many_workgroups_per_cu(K):
    """An odd number of 32-wide K-steps ends on the peeled last step, whose A-fragment reads (LDS buffer 0) must be complete in
    EVERY wave before the epilogue overwrites that memory with the transposed accumulators (the barrier was
    missing; the model's own shapes all have an even K / 32).  Many small tiles per CU and repeated launches give the waves
    of a workgroup every chance to drift apart; every piece count / tile shape the engine issues."""
    M, N = 8192, 512
    A, W, b = rnd(M, K, seed=11), rnd(N, K, seed=12, scale=K ** -0.5), rnd(N, seed=13)
    ref = A.double() @ W.double().t() + b.double()
    aw = A.double().abs() @ W.double().abs().t() + b.double().abs()
    Ad, bd = dev(A), dev(b)
    for name, Wp, rel in (("bf16x3", hip.pack_w_bf16x3(dev(W)), 2.0 ** -15), ("bf16x6", hip.pack_w_bf16x6(dev(W)), 2.0 ** -21),
                          ("f16x3", hip.pack_w_f16x3(dev(W)), 2.0 ** -19)):
        outs = []
        for rep in range(6):
            C = torch.full((M, N), float("nan"), device=DEV)
            hip.gemm(Ad, Wp, C, M=M, N=N, K=K, bias=bd)
            outs.append(C)
        torch.cuda.synchronize()
        err = (outs[0].cpu().double() - ref).abs()
        assert bool((err <= aw * rel + 1e-6).all()), f"{name} K={K}: worst {float((err / (aw * rel + 1e-6)).max()):.2f} of the bound"
        for C in outs[1:]:
            assert torch.equal(C, outs[0]), f"{name} K={K}: repeated launches differ (a race in the kernel)"


@pytest.mark.parametrize("M,N,K", [(300, 200, 96), (129, 64, 64), (70, 33, 192), (257, 2048, 256), (640, 384, 1536),
                                   (6, 768, 384), (12, 2048, 256), (16, 1536, 384), (64, 384, 128), (1, 64, 64), (33, 96, 320)])
def test_gemm_bf16x6_is_fp32_class(M, N, K):
    """Six passes over three bf16 pieces per operand: errors of the fp32-MFMA kernel's size (bound: 2^-21 |A||W|)."""
    A, W, b = rnd(M, K, seed=1), rnd(N, K, seed=2, scale=K ** -0.5), rnd(N, seed=3)
    C = torch.full((M, N), float("nan"), device=DEV)
    hip.gemm(dev(A), hip.pack_w_bf16x6(dev(W)), C, M=M, N=N, K=K, bias=dev(b))
    torch.cuda.synchronize()
    ref = A.double() @ W.double().t() + b.double()
    bound = (A.double().abs() @ W.double().abs().t() + b.double().abs()) * 2.0 ** -21
    err = (C.cpu().double() - ref).abs()
    assert bool((err <= bound).all()), f"{M}x{N}x{K}: worst {float((err / bound).max()):.2f} of the bound"


def test_gemm_bf16x6_epilogues_and_addvec():
    M, N, K = 200, 256, 128
    A, W, b = rnd(M, K, seed=4), rnd(N, K, seed=5, scale=K ** -0.5), rnd(N, seed=6)
    R, pv = rnd(M, N, seed=7), rnd(K, seed=9)
    Wp = hip.pack_w_bf16x6(dev(W))
    ref = A @ W.t() + b
    C = torch.empty(M, N, device=DEV)
    hip.gemm(dev(A), Wp, C, M=M, N=N, K=K, bias=dev(b), epilogue=hip.EPI_GELU)
    close(C, F.gelu(ref), 3e-5, "gelu")
    hip.gemm(dev(A), Wp, C, M=M, N=N, K=K, bias=dev(b), epilogue=hip.EPI_RES, R=dev(R))
    close(C, R + ref, 5e-5, "res")
    hip.gemm(dev(A), Wp, C, M=M, N=N, K=K, bias=dev(b), prologue=hip.PRO_ADDVEC, pro_vec=dev(pv))
    close(C, (A + pv) @ W.t() + b, 5e-5, "addvec prologue")
    wg, bg = pack.pack_glu(W, b)
    G = torch.empty(M, N // 2, device=DEV)
    hip.gemm(dev(A), hip.pack_w_bf16x6(dev(wg)), G, M=M, N=N, K=K, bias=dev(bg), epilogue=hip.EPI_GLU)
    close(G, ref[:, : N // 2] * torch.sigmoid(ref[:, N // 2:]), 3e-5, "glu")
    with pytest.raises(hip.SoproHipError):
        hip.gemm(dev(A), Wp, C, M=M, N=N, K=K, prologue=hip.PRO_ELU)


@pytest.mark.parametrize("pieces", [3, 1])
@pytest.mark.parametrize("M,heads,V,K", [(200, 3, 2048, 256), (6, 16, 2048, 256), (77, 2, 128, 64)])
def test_gemm_argmax_epilogue_equals_argmax_of_the_logits(pieces, M, heads, V, K):
    """c_mode 5: the per-head arg-max is taken in the contraction's epilogue (per 64-column tile) and finished by
    sopro_argmax_partials_i32 - the same tokens as writing the logits and running the arg-max kernel over them, exact ties
    (duplicated weight rows) going to the lower index like torch.argmax (src/sopro/model.py:338-345)."""
    N = heads * V
    A, W, b = rnd(M, K, seed=61), rnd(N, K, seed=62, scale=K ** -0.5), rnd(N, seed=63, scale=0.1)
    twin = 900 if V > 900 else 100
    W[twin] = W[5]
    b[twin] = b[5]  # an exact tie inside head 0: the lower index (5) must win where it is the maximum of a row
    A[0] = 8.0 * W[5] / W[5].norm()  # and make it the maximum of row 0
    Wp = hip.pack_w_bf16(dev(W), pieces)
    logits = torch.empty(M, N, device=DEV)
    hip.gemm(dev(A), Wp, logits, M=M, N=N, K=K, bias=dev(b))
    want = torch.full((M, heads + 2), -1, dtype=torch.int32, device=DEV)
    hip.argmax_rows(logits, want, rows=M * heads, N=V, ldo=heads + 2, o_off=1, inner=heads)
    part = torch.full((M, N // 64 + 3, 2), float("nan"), device=DEV)
    got = torch.full((M, heads + 2), -1, dtype=torch.int32, device=DEV)
    hip.gemm(dev(A), Wp, None, M=M, N=N, K=K, bias=dev(b), c_mode=5, C2=part, ldc2=N // 64 + 3)
    hip.argmax_partials(part, got, rows=M, heads=heads, per_head=V // 64, V=V, ldp=N // 64 + 3, ldo=heads + 2, o_off=1)
    torch.cuda.synchronize()
    assert torch.equal(got, want)
    assert int(got[0, 1]) == 5 and bool((got[:, 0] == -1).all()) and bool((got[:, -1] == -1).all())
    ref = logits.cpu().view(M, heads, V).argmax(-1).to(torch.int32)
    assert torch.equal(got[:, 1:-1].cpu(), ref)


@pytest.mark.parametrize("heads,per_head", [(1, 1), (3, 5), (4, 8), (2, 37), (31, 32)])
def test_argmax_partials_order_free_reduction(heads, per_head):
    """Round 6: sopro_argmax_partials_i32 reduces a head's (value, column) pairs with eight lanes per head (strided walk + three xor
    exchanges) instead of one thread walking them in order: the winner - largest value, smallest column among equals - must not depend
    on which lane saw which pair.  Synthetic pairs: ties between pairs of different lanes, heads shorter than eight pairs, rows whose
    values are all -inf (every pair ties), padded rows (ldp > heads * per_head, ldo > heads)."""
    import numpy as np
    V, rows = 64 * per_head, 203
    g = np.random.default_rng(7 + heads * 100 + per_head)
    val = g.standard_normal((rows, heads, per_head)).astype(np.float32)
    col = np.stack([np.stack([64 * j + g.integers(0, 64, rows) for j in range(per_head)], -1) + h * V for h in range(heads)], 1).astype(np.int32)
    for r in range(0, rows, 3):  # exact ties: the maximum appears in two or three pairs (different lanes when per_head > 1)
        for h in range(heads):
            js = g.choice(per_head, size=min(per_head, 1 + (r % 3)), replace=False)
            val[r, h, js] = 9.0
    val[5] = -np.inf
    ldp, ldo = heads * per_head + 3, heads + 2
    part = torch.full((rows, ldp, 2), float("nan"))
    part[:, :heads * per_head, 0] = torch.from_numpy(val.reshape(rows, -1))
    part[:, :heads * per_head, 1] = torch.from_numpy(col.reshape(rows, -1)).view(torch.float32)
    got = torch.full((rows, ldo), -1, dtype=torch.int32, device=DEV)
    hip.argmax_partials(dev(part), got, rows=rows, heads=heads, per_head=per_head, V=V, ldp=ldp, ldo=ldo, o_off=1)
    torch.cuda.synchronize()
    want = np.zeros((rows, heads), np.int32)
    for r in range(rows):
        for h in range(heads):
            m = val[r, h].max()
            want[r, h] = col[r, h][val[r, h] == m].min() - h * V  # (a row of -inf: every pair ties - the smallest column)
    assert np.array_equal(got[:, 1:-1].cpu().numpy(), want)
    assert bool((got[:, 0] == -1).all()) and bool((got[:, -1] == -1).all())


@pytest.mark.parametrize("M,K", [(6, 384), (200, 384), (1000, 384), (70, 64), (1, 32)])
def test_gemm_bf16x6_fused_rmsnorm(M, K):
    """RMSNorm inside the GEMM (ext.rms_norm; reference: src/sopro/nn/blocks.py:26-37 followed by the block's GLU / FF1
    contraction): the norm's weight vector is folded into W, the row scale is applied to the accumulator."""
    N, eps = 256, 1e-6
    X, W, b, nw = rnd(M, K, seed=51, scale=3.0), rnd(N, K, seed=52, scale=K ** -0.5), rnd(N, seed=53), 1.0 + 0.3 * rnd(K, seed=54)
    X[M // 2] *= 1e-3  # rows of very different energy
    xn = (X.double() * torch.rsqrt((X.double() ** 2).mean(-1, keepdim=True) + eps)) * nw.double()
    ref = (xn @ W.double().t() + b.double()).float()
    Wf = (W * nw[None, :]).contiguous()
    C = torch.full((M, N), float("nan"), device=DEV)
    hip.gemm(dev(X), hip.pack_w_bf16x6(dev(Wf)), C, M=M, N=N, K=K, bias=dev(b), rms_eps=eps)
    close(C, ref, 2e-5, "rmsnorm+gemm")
    hip.gemm(dev(X), hip.pack_w_bf16x6(dev(Wf)), C, M=M, N=N, K=K, bias=dev(b), epilogue=hip.EPI_GELU, rms_eps=eps)
    close(C, F.gelu(ref), 2e-5, "rmsnorm+gemm+gelu")
    wg, bg = pack.pack_glu(Wf, b)
    G = torch.full((M, N // 2), float("nan"), device=DEV)
    hip.gemm(dev(X), hip.pack_w_bf16x6(dev(wg)), G, M=M, N=N, K=K, bias=dev(bg), epilogue=hip.EPI_GLU, rms_eps=eps)
    close(G, ref[:, : N // 2] * torch.sigmoid(ref[:, N // 2:]), 2e-5, "rmsnorm+gemm+glu")
    # the unfused pair of launches computes the same thing
    nrm = torch.empty(M, K, device=DEV)
    hip.norm(dev(X), nrm, dev(nw), rows=M, C_=K, eps=eps)
    C2 = torch.empty(M, N, device=DEV)
    hip.gemm(nrm, hip.pack_w_bf16x6(dev(W)), C2, M=M, N=N, K=K, bias=dev(b))
    hip.gemm(dev(X), hip.pack_w_bf16x6(dev(Wf)), C, M=M, N=N, K=K, bias=dev(b), rms_eps=eps)
    close(C, C2.cpu(), 2e-5, "fused vs unfused")
    with pytest.raises(hip.SoproHipError):  # a two-piece weight has no fused norm
        hip.gemm(dev(X), hip.pack_w_bf16x3(dev(Wf)), C, M=M, N=N, K=K, rms_eps=eps)


@pytest.mark.parametrize("M,N,K", [(300, 200, 96), (129, 64, 64), (6400, 384, 1536), (6400, 1536, 384), (6, 768, 384), (12, 2048, 1024), (1, 64, 1536), (33, 96, 320)])
def test_gemm_f16x3_is_fp32_class(M, N, K):
    """Two fp16 pieces per operand, three MFMA passes (round 3: the NAR contractions): 22 mantissa bits, so the error against
    fp64 must stay within a few 2^-22 of |A||W| - the class of the six-pass bf16 form (2^-24) and of an fp32 fma chain - over
    the magnitudes of the NAR stream (rows of RMS 0.1 .. 10, weights of very different size in one matrix: the power-of-two
    scales must not cost precision), with every form NAR uses."""
    g = torch.Generator().manual_seed(7)
    A = rnd(M, K, seed=71) * torch.exp(0.8 * torch.randn(M, 1, generator=g)).clamp(0.1, 10.0)
    W, b = rnd(N, K, seed=72, scale=K ** -0.5), rnd(N, seed=73)
    W[: N // 4] *= 1e-3  # small and large weights in one matrix
    Wp = hip.pack_w_f16x3(dev(W))
    ref = A.double() @ W.double().t() + b.double()
    aw = A.double().abs() @ W.double().abs().t()
    # 2^-20 of |A||W| (22-bit operands + fp32 accumulation) + the absolute resolution of a scaled fp16 piece (2^-25 / 8 per activation)
    bound = aw * 2.0 ** -20 + W.double().abs().sum(1)[None, :] * 4e-9 + b.double().abs() * 2.0 ** -22 + 1e-30
    C = torch.full((M, N), float("nan"), device=DEV)
    hip.gemm(dev(A), Wp, C, M=M, N=N, K=K, bias=dev(b))
    torch.cuda.synchronize()
    err = (C.cpu().double() - ref).abs()
    assert bool(torch.isfinite(C).all()) and bool((err <= bound).all()), f"{M}x{N}x{K}: worst {float((err / bound).max()):.2f} of the bound"
    assert float((err / (aw + b.double().abs())).mean()) < 2.0 ** -23  # typical error: a fraction of the bound (measured 1.3e-8: tools/f16x3_probe.py)
    # outside the domain: a low-energy row keeps the absolute floor, a huge one saturates instead of producing inf / NaN
    A2 = A.clone()
    A2[0] *= 1e-3
    if M > 1:
        A2[1] = 2.0e4
    hip.gemm(dev(A2), Wp, C, M=M, N=N, K=K, bias=dev(b))
    torch.cuda.synchronize()
    assert bool(torch.isfinite(C).all())
    e0 = (C[0].cpu().double() - (A2[0].double() @ W.double().t() + b.double())).abs()
    assert bool((e0 <= (A2[0].double().abs() @ W.double().abs().t()) * 2.0 ** -20 + W.double().abs().sum(1) * 4e-9 + b.double().abs() * 2.0 ** -22).all())
    # the forms: GELU, residual, head-id prologue, fused RMSNorm, GLU, arg-max partials
    R, pv = rnd(M, N, seed=74), rnd(K, seed=75)
    hip.gemm(dev(A), Wp, C, M=M, N=N, K=K, bias=dev(b), epilogue=hip.EPI_GELU)
    close(C, F.gelu(ref.float()), 3e-5 * max(1.0, float(ref.abs().max())), "gelu")
    hip.gemm(dev(A), Wp, C, M=M, N=N, K=K, bias=dev(b), epilogue=hip.EPI_RES, R=dev(R))
    close(C, R + ref.float(), 3e-5 * max(1.0, float(ref.abs().max())), "res")
    hip.gemm(dev(A), Wp, C, M=M, N=N, K=K, bias=dev(b), prologue=hip.PRO_ADDVEC, pro_vec=dev(pv))
    close(C, ((A + pv).double() @ W.double().t() + b.double()).float(), 3e-5 * max(1.0, float(ref.abs().max())), "addvec")
    if K % 32 == 0:
        eps, nw = 1e-6, 1.0 + 0.3 * rnd(K, seed=76)
        Wf = (W * nw[None, :]).contiguous()
        xn = (A.double() * torch.rsqrt((A.double() ** 2).mean(-1, keepdim=True) + eps)) * nw.double()
        refn = (xn @ W.double().t() + b.double()).float()
        Wfp = hip.pack_w_f16x3(dev(Wf))
        hip.gemm(dev(A), Wfp, C, M=M, N=N, K=K, bias=dev(b), rms_eps=eps)
        close(C, refn, 2e-5 * max(1.0, float(refn.abs().max())), "fused rmsnorm")
        if N % 64 == 0:
            wg, bg = pack.pack_glu(Wf, b)
            G = torch.full((M, N // 2), float("nan"), device=DEV)
            hip.gemm(dev(A), hip.pack_w_f16x3(dev(wg)), G, M=M, N=N, K=K, bias=dev(bg), epilogue=hip.EPI_GLU, rms_eps=eps)
            close(G, refn[:, : N // 2] * torch.sigmoid(refn[:, N // 2:]), 2e-5 * max(1.0, float(refn.abs().max())), "rmsnorm + glu")
    if N % 64 == 0:
        part = torch.full((M, N // 64, 2), float("nan"), device=DEV)
        got = torch.full((M, 1), -1, dtype=torch.int32, device=DEV)
        hip.gemm(dev(A), Wp, None, M=M, N=N, K=K, bias=dev(b), c_mode=5, C2=part, ldc2=N // 64)
        hip.argmax_partials(part, got, rows=M, heads=1, per_head=N // 64, V=N, ldp=N // 64, ldo=1, o_off=0)
        hip.gemm(dev(A), Wp, C, M=M, N=N, K=K, bias=dev(b))
        torch.cuda.synchronize()
        assert torch.equal(got[:, 0].cpu().long(), C.cpu().argmax(-1))


def test_fused_rmsnorm_forms_do_not_depend_on_the_tile_shape():
    """Round 6: the f16 three-pass forms with a fused RMSNorm (plain, GELU, GLU) and the plain GLU on the three tile shapes the engine
    picks by row count (override 1 = 128 x 128 for whole passes, 4 = 64 x 128, 5 = 64 x 64 / 64 x 128 for streaming chunks and small
    batches): bit-identical.  They were not before round 6 - `acc * row_scale + bias` was contracted into a fused multiply-add in some
    instantiations and not in others (1 ulp apart), so a refinement over a whole pass and the same rows in a small batch could differ;
    the epilogue now rounds the row-scaled product explicitly (gemm_epilogue.h).  The six-pass form (the refinement's fallback) likewise."""
    M, K = 1100, 384  # ragged last row tile on every shape
    lib = hip.load()
    for name, N, kw, pack in (("rms plain", 256, {}, hip.pack_w_f16x3), ("rms gelu", 640, dict(epilogue=hip.EPI_GELU), hip.pack_w_f16x3),
                              ("rms glu", 768, dict(epilogue=hip.EPI_GLU), hip.pack_w_f16x3), ("glu", 768, dict(epilogue=hip.EPI_GLU, rms_eps=0.0), hip.pack_w_f16x3),
                              ("six-pass rms gelu", 512, dict(epilogue=hip.EPI_GELU), hip.pack_w_bf16x6)):
        A, W, b = rnd(M, K, seed=171), rnd(N, K, seed=172, scale=K ** -0.5), rnd(N, seed=173)
        Ad, bd, Wp = dev(A), dev(b), pack(dev(W))
        outs = []
        try:
            for cfg in (1, 4, 5):
                lib.sopro_gemm_bf16_set_tile_override(cfg)
                C = torch.full((M, N // 2 if kw.get("epilogue") == hip.EPI_GLU else N), float("nan"), device=DEV)
                hip.gemm(Ad, Wp, C, M=M, N=N, K=K, bias=bd, **dict(dict(rms_eps=1e-6), **kw))
                torch.cuda.synchronize()
                outs.append(C.cpu())
        finally:
            lib.sopro_gemm_bf16_set_tile_override(0)
        assert bool(torch.isfinite(outs[0]).all()), name
        assert torch.equal(outs[0], outs[1]) and torch.equal(outs[0], outs[2]), name


def test_gemm_eight_wave_tiles_are_the_same_function():
    """128x128 tiles on eight waves (round 4 developer form of the three-pass decoder contraction, tile override 7: four waves per
    SIMD; measured faster alone, not in the pipeline - profiles/r04_experiments.md) against the four-wave form: every output element
    runs the same K loop, so every form of that path is bit-identical - plain, GELU, residual, activated copies, ELU prologue -
    with a ragged last row tile and a K that ends in an odd step."""
    M, N, K = 1300, 640, 416  # ragged last row tile, 5 column tiles, 13 K-steps
    A, W, b, R = rnd(M, K, seed=81), rnd(N, K, seed=82, scale=K ** -0.5), rnd(N, seed=83), rnd(M, N, seed=84)
    Ad, bd, Rd = dev(A), dev(b), dev(R)
    lib, Wp = hip.load(), hip.pack_w_bf16x3(dev(W))

    def both(fn):
        outs = []
        try:
            for cfg in (1, 7):
                lib.sopro_gemm_bf16_set_tile_override(cfg)
                C = torch.full((M, N), float("nan"), device=DEV)
                extra = fn(C)
                torch.cuda.synchronize()
                outs.append([C.cpu()] + [e.cpu() for e in (extra or [])])
        finally:
            lib.sopro_gemm_bf16_set_tile_override(0)
        for x, y in zip(*outs):
            assert bool(torch.isfinite(x).all()) and torch.equal(x, y)

    both(lambda C: hip.gemm(Ad, Wp, C, M=M, N=N, K=K, bias=bd))
    both(lambda C: hip.gemm(Ad, Wp, C, M=M, N=N, K=K, bias=bd, epilogue=hip.EPI_GELU))
    both(lambda C: hip.gemm(Ad, Wp, C, M=M, N=N, K=K, bias=bd, epilogue=hip.EPI_RES, R=Rd))
    both(lambda C: hip.gemm(Ad, Wp, C, M=M, N=N, K=K, bias=bd, c_mode=3))
    both(lambda C: hip.gemm(Ad, Wp, C, M=M, N=N, K=K, bias=bd, prologue=hip.PRO_ELU))

    def raw_and_act(C):
        C2 = torch.full((M, N), float("nan"), device=DEV)
        hip.gemm(Ad, Wp, C, M=M, N=N, K=K, bias=bd, c_mode=4, C2=C2)
        return [C2]
    both(raw_and_act)


@pytest.mark.parametrize("pieces,M", [(2, 3 * 400), (1, 3 * 400), (2, 2 * 6)])
def test_gemm_rope_epilogue_equals_the_contraction_followed_by_rope(pieces, M):
    """Round 5: SOPRO_EPI_ROPE rotates the q | k heads of a fused qkv projection in the contraction's epilogue (the tile is in LDS, a
    column's partner is at hand) instead of a pass of sopro_rope_f32 over C (HF:modeling_mimi.py:511-566).  Same arithmetic: equal to
    contraction + rope up to the contraction of a multiply-add (1 ulp class), on 128x128 tiles (many rows), 64x64 tiles and split-K
    (a streaming chunk's few rows), three-pass and one-pass operands; the v block is left alone; positions restart per utterance."""
    from sopro_amd.pack import rope_tables

    HS, H, dh, K = 512, 8, 64, 512
    N = 3 * HS
    n = M // (3 if M >= 1200 else 2)  # rows per utterance
    A, W, b = rnd(M, K, seed=71), rnd(N, K, seed=72, scale=K ** -0.5), rnd(N, seed=73)
    cos_t, sin_t = (t.to(DEV) for t in rope_tables(1024, dh, 10000.0))
    Ad, bd = dev(A), dev(b)
    Wp = hip.pack_w_bf16(dev(W), pieces)
    pos0 = 37
    want = torch.empty(M, N, device=DEV)
    hip.gemm(Ad, Wp, want, M=M, N=N, K=K, bias=bd)
    v_before = want[:, 2 * HS:].clone()
    hip.rope(want, cos_t, sin_t, rows=M, rows_per_seg=n, pos0=pos0, H=2 * H, dh=dh, ldx=N)
    got = torch.full((M, N), float("nan"), device=DEV)
    hip.gemm(Ad, Wp, got, M=M, N=N, K=K, bias=bd, epilogue=hip.EPI_ROPE, rope=(cos_t, sin_t, 2 * HS, dh, pos0, n))
    torch.cuda.synchronize()
    assert bool(torch.isfinite(got).all())
    assert torch.equal(got[:, 2 * HS:], v_before)  # v: untouched
    err = float((got - want).abs().max())
    assert err <= 4e-6 * float(want.abs().max()), err
    # ... and it is a rotation: norms of (e, e + dh/2) pairs are those of the unrotated projection
    raw = torch.empty(M, N, device=DEV)
    hip.gemm(Ad, Wp, raw, M=M, N=N, K=K, bias=bd)
    pr = lambda t: (t[:, :2 * HS].reshape(M, 2 * H, 2, dh // 2) ** 2).sum(2)
    assert float((pr(got) - pr(raw)).abs().max()) <= 1e-4 * float(pr(raw).max())
    with pytest.raises(hip.SoproHipError):  # heads must be whole
        hip.gemm(Ad, Wp, got, M=M, N=N, K=K, bias=bd, epilogue=hip.EPI_ROPE, rope=(cos_t, sin_t, 2 * HS + 8, dh, pos0, n))


@pytest.mark.parametrize("pieces,M,n", [(2, 3 * 413, 413), (1, 3 * 413, 413), (2, 2 * 6, 6)])
def test_gemm_fused_layernorm_equals_norm_followed_by_the_contraction(pieces, M, n):
    """Round 5 (sopro_gemm_split_ext.ln_stats / ln_stats_out): the pre-norms of the Mimi decoder transformer (HF:modeling_mimi.py:796-833)
    ride on the contractions either side of them.  (1) sopro_row_stats_f32's (mean, squared deviations) per 64 columns are those of the
    float64 statistics; (2) an EPI_RES contraction with ln_stats_out leaves BIT-IDENTICAL pairs for the stream it wrote (same arithmetic,
    same order), on 128x128 tiles, 64x64 tiles and split-K, with the stream in padded segments as the engine holds it; (3) a contraction
    with ln_stats stages (x - mean) * rstd: equal to sopro_norm_f32 (LN) followed by the plain contraction within fp32 round-off, for a
    stream with a large common offset (mean 30, spread 1: E[x^2] - E[x]^2 would lose 3 digits) and per-row scales over 1e-3 .. 1e3."""
    HS, N2, eps, PAD = 512, 1536, 1e-5, 2
    rng = np.random.default_rng(5)
    xs = (PAD + n) * HS  # one utterance of the padded stream
    B = M // n
    X0 = torch.from_numpy(rng.standard_normal((B, PAD + n, HS)).astype(np.float32))
    X0 = X0 * torch.from_numpy(np.logspace(-3, 3, B * (PAD + n)).astype(np.float32)).reshape(B, PAD + n, 1) + 30.0
    Xd = dev(X0)
    rows = Xd[:, PAD:].reshape(M, HS)
    st = torch.full((M, HS // 64, 2), float("nan"), device=DEV)
    hip.row_stats(Xd, M, HS, st, x_seg_stride=xs, rows_per_seg=n, x_off=PAD * HS)
    g64 = rows.double().reshape(M, HS // 64, 64)
    mag = g64.abs().amax(-1)  # fp32 sums of 64 values: errors are relative to the group's largest magnitude
    assert bool(((st[..., 0].double() - g64.mean(-1)).abs() <= 4e-7 * mag).all())
    m2 = ((g64 - g64.mean(-1, keepdim=True)) ** 2).sum(-1)
    assert bool(((st[..., 1].double() - m2).abs() <= 1e-5 * m2 + 1e-6 * mag * mag).all())
    # (2) producer: X <- X + scale * (A Wo^T), statistics of the result
    K = 512
    A, Wo, sc = rnd(M, K, seed=81), rnd(HS, K, seed=82, scale=K ** -0.5), rnd(HS, seed=83)
    Wop = hip.pack_w_bf16(dev(Wo), pieces)
    st_out = torch.full((M, HS // 64, 2), float("nan"), device=DEV)
    hip.gemm(dev(A), Wop, Xd, M=M, N=HS, K=K, epilogue=hip.EPI_RES, R=Xd, scale=dev(sc), rows_per_seg=n, c_seg_stride=xs, r_seg_stride=xs,
             c_off=PAD * HS, r_off=PAD * HS, ln_stats_out=st_out)
    hip.row_stats(Xd, M, HS, st, x_seg_stride=xs, rows_per_seg=n, x_off=PAD * HS)
    torch.cuda.synchronize()
    assert torch.equal(st_out, st)
    # (3) consumer
    lnw, lnb = rnd(HS, seed=84) * 0.3 + 1.0, rnd(HS, seed=85) * 0.2
    W = rnd(N2, HS, seed=86, scale=HS ** -0.5)
    y = torch.empty(M, HS, device=DEV)
    hip.norm(Xd, y, dev(lnw), rows=M, C_=HS, eps=eps, kind=hip.NORM_LN, b=dev(lnb), rows_per_seg=n, x_seg_stride=xs, x_off=PAD * HS)
    want = torch.empty(M, N2, device=DEV)
    hip.gemm(y, hip.pack_w_bf16(dev(W), pieces), want, M=M, N=N2, K=HS, epilogue=hip.EPI_GELU)
    Wf = hip.pack_w_bf16(dev(W * lnw[None, :]), pieces)
    bf = dev((W.double() @ lnb.double()).float())
    got = torch.full((M, N2), float("nan"), device=DEV)
    hip.gemm(Xd, Wf, got, M=M, N=N2, K=HS, bias=bf, epilogue=hip.EPI_GELU, rows_per_seg=n, a_seg_stride=xs, a_off=PAD * HS, ln_stats=st_out, ln_eps=eps)
    torch.cuda.synchronize()
    assert bool(torch.isfinite(got).all())
    rows = Xd[:, PAD:].reshape(M, HS).double().cpu()
    ref = torch.nn.functional.gelu(torch.nn.functional.layer_norm(rows, (HS,), lnw.double(), lnb.double(), eps) @ W.double().T).to(DEV)
    tol = (8e-3 if pieces == 1 else 2e-5) * float(ref.abs().max())
    assert float((got.double() - ref).abs().max()) <= tol and float((want.double() - ref).abs().max()) <= tol
    assert float((got - want).abs().max()) <= tol
    with pytest.raises(hip.SoproHipError):  # the statistics are per 64 columns
        hip.gemm(Xd, hip.pack_w_bf16(dev(W[:, :480].contiguous()), pieces), got, M=M, N=N2, K=480, ln_stats=st_out, ln_eps=eps)


@pytest.mark.parametrize("K", [256, 384, 512])
def test_gemm_activation_stationary_form_is_the_same_function(K):
    """Round 5 (csrc/gemm_astat.hip): short-K contractions with the A block resident in LDS and barrier-free column-tile walks, tile
    override 9, against the 128x128 tile kernel (override 1): the same products in the same order per output element, so plain /
    GELU / residual (+ layer scale, in place) forms are bit-identical - with a ragged last row block, a ragged last column tile
    (N % 256 != 0, N % 32 != 0), several column tiles per workgroup, segments with overlapping rows (lda < K: a convolution window)
    - and both within fp32-fma class of the float64 product."""
    lib = hip.load()
    if not (lib.sopro_build_flags() & 1):
        pytest.skip("the activation-stationary form is a measured no-go: compiled into the developer build only (make DEV=1, SOPRO_HIP_LIB)")
    M, N = 4200, 1156  # 66 row blocks (the last one 40 rows), 5 column tiles (the last one 132 columns: 4 full + 1 ragged 32-tile)
    A, W, b, R, sc = rnd(M, K, seed=91), rnd(N, K, seed=92, scale=K ** -0.5), rnd(N, seed=93), rnd(M, N, seed=94), rnd(N, seed=95)
    Ad, bd, Rd, scd = dev(A), dev(b), dev(R), dev(sc)
    Wp = hip.pack_w_bf16x3(dev(W))
    ref = A.double() @ W.double().t() + b.double()
    mag = A.double().abs() @ W.double().abs().t() + b.double().abs()

    def both(fn, check=None):
        outs = []
        try:
            for cfg in (1, 9):
                lib.sopro_gemm_bf16_set_tile_override(cfg)
                outs.append(fn().cpu())
        finally:
            lib.sopro_gemm_bf16_set_tile_override(0)
        assert bool(torch.isfinite(outs[0]).all()) and torch.equal(outs[0], outs[1])
        if check is not None:
            check(outs[1])

    def plain():
        C = torch.full((M, N), float("nan"), device=DEV)
        hip.gemm(Ad, Wp, C, M=M, N=N, K=K, bias=bd)
        torch.cuda.synchronize()
        return C
    both(plain, lambda C: (lambda e: e < 2e-5)(float(((C.double() - ref).abs() / mag).max())) or pytest.fail("three-pass error class"))

    def gelu():
        C = torch.full((M, N), float("nan"), device=DEV)
        hip.gemm(Ad, Wp, C, M=M, N=N, K=K, bias=bd, epilogue=hip.EPI_GELU)
        torch.cuda.synchronize()
        return C
    both(gelu)

    def res_in_place():  # x <- x + scale * (A W^T + b): R aliases C
        C = Rd.clone()
        hip.gemm(Ad, Wp, C, M=M, N=N, K=K, bias=bd, epilogue=hip.EPI_RES, R=C, scale=scd)
        torch.cuda.synchronize()
        return C
    both(res_in_place, lambda C: (lambda e: e < 1e-4)(float((C.double() - (R.double() + sc.double() * ref)).abs().max())) or pytest.fail("residual form"))

    # segments of overlapping rows: 3 utterances x (2 zero rows + 700 rows) of K / 2 channels, A-row t = [x[t-1] | x[t]] (lda = K / 2)
    Cin, T, B = K // 2, 700, 3
    xs = torch.zeros(B, 1 + T, Cin)
    xs[:, 1:] = rnd(B, T, Cin, seed=96)
    xd = dev(xs)
    N2 = 320

    def conv_window():
        C = torch.full((B, T, N2), float("nan"), device=DEV)
        hip.gemm(xd, hip.pack_w_bf16x3(dev(W[:N2])), C, M=B * T, N=N2, K=K, lda=Cin, rows_per_seg=T, a_seg_stride=(1 + T) * Cin, c_seg_stride=T * N2)
        torch.cuda.synchronize()
        return C
    lib.sopro_gemm_bf16_set_tile_override(9)
    try:
        got = conv_window().cpu()
    finally:
        lib.sopro_gemm_bf16_set_tile_override(0)
    win = torch.cat([xs[:, :-1], xs[:, 1:]], dim=-1)  # [B, T, K]
    want = win.double() @ W[:N2].double().t()
    assert float((got.double() - want).abs().max()) < 1e-4
    both(conv_window)


@pytest.mark.parametrize("pieces,M,N,K,epi", [(3, 6, 384, 1536, "res"), (3, 6, 768, 1152, "glu"), (3, 12, 2048, 1024, "none"), (2, 12, 512, 2048, "res"),
                                              (2, 12, 1024, 3584, "none"), (2, 33, 4096, 2048, "none"), (3, 1, 64, 1536, "gelu")])
def test_gemm_split_k_small_m_is_exact_class_and_deterministic(pieces, M, N, K, epi):
    """Few-row problems run split-K (last-arriver reduction in slice order): same accuracy class, bitwise repeatable."""
    assert hip._auto_ksplit(M, N, K, pieces, hip.EPI_GLU if epi == "glu" else hip.EPI_NONE) > 1
    A, W, b, R = rnd(M, K, seed=41), rnd(N, K, seed=42, scale=K ** -0.5), rnd(N, seed=43), rnd(M, N, seed=44)
    pre = A.double() @ W.double().t() + b.double()
    mag = A.double().abs() @ W.double().abs().t() + b.double().abs()
    kw, Wd, bd, nout = {}, W, b, N
    if epi == "res":
        ref, kw = pre + R.double(), dict(epilogue=hip.EPI_RES, R=dev(R))
    elif epi == "gelu":
        ref, kw = F.gelu(pre), dict(epilogue=hip.EPI_GELU)
    elif epi == "glu":
        Wd, bd = pack.pack_glu(W, b)
        ref, mag, nout, kw = pre[:, : N // 2] * torch.sigmoid(pre[:, N // 2:]), mag[:, : N // 2], N // 2, dict(epilogue=hip.EPI_GLU)
    else:
        ref = pre
    Wp = hip.pack_w_bf16(dev(Wd), pieces)
    outs = []
    for _ in range(3):
        C = torch.full((M, nout), float("nan"), device=DEV)
        hip.gemm(dev(A), Wp, C, M=M, N=N, K=K, bias=dev(bd), **kw)
        outs.append(C.cpu())
    assert torch.equal(outs[0], outs[1]) and torch.equal(outs[0], outs[2])
    bound = mag * (2.0 ** -21 if pieces == 3 else 2.0 ** -15) + 1e-6
    err = (outs[0].double() - ref).abs()
    assert bool((err <= bound).all()), f"worst {float((err / bound).max()):.2f} of the bound"
    # two streams at once: each has its own split-K scratch
    s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
    C1, C2 = torch.empty(M, nout, device=DEV), torch.empty(M, nout, device=DEV)
    Ad, bdv = dev(A), dev(bd)
    torch.cuda.synchronize()
    for _ in range(5):
        with torch.cuda.stream(s1):
            hip.gemm(Ad, Wp, C1, M=M, N=N, K=K, bias=bdv, **kw)
        with torch.cuda.stream(s2):
            hip.gemm(Ad, Wp, C2, M=M, N=N, K=K, bias=bdv, **kw)
    torch.cuda.synchronize()
    assert torch.equal(C1.cpu(), outs[0]) and torch.equal(C2.cpu(), outs[0])


def _planes(t, P):
    """Split-form rows [.., P floats] (every 32 channels = [32 hi | 32 lo] bf16) -> the fp32 values hi + lo they encode."""
    b = t.contiguous().view(torch.bfloat16).view(*t.shape[:-1], P // 32, 2, 32).float()
    return (b[..., 0, :] + b[..., 1, :]).reshape(*t.shape[:-1], P)


def test_gemm_bf16x3_split_plane_producer_and_consumer():
    """c_mode 1 / 2 write ELU(C) in split form; a_split consumes it through overlapping row windows."""
    B, T, ci, co, r = 2, 19, 64, 64, 4
    x = rnd(B, T, ci, seed=21)
    wt, bt = rnd(ci, co, 2 * r, seed=22, scale=(2 * ci) ** -0.5), rnd(co, seed=23)
    wp, bp = pack.pack_convtr1d(wt, bt, r)
    y = F.conv_transpose1d(x.transpose(1, 2), wt, bt, stride=r)
    raw_ref = y[..., : y.shape[-1] - r].transpose(1, 2).contiguous()  # [B, T*r, co]
    buf = torch.zeros(B, 1 + T, ci)
    buf[:, 1:] = x
    raw = torch.zeros(B, 2 + T * r, co, device=DEV)
    act = torch.zeros(B, 2 + T * r, co, device=DEV)
    hip.gemm(dev(buf), hip.pack_w_bf16x3(dev(wp)), raw, M=B * T, N=r * co, K=2 * ci, lda=ci, bias=dev(bp), rows_per_seg=T,
             a_seg_stride=(1 + T) * ci, c_off=2 * co, c_seg_stride=(2 + T * r) * co, ldc=r * co, c_mode=2, C2=act,
             ldc2=r * co, c2_seg_stride=(2 + T * r) * co, c2_off=2 * co)
    close(raw[:, 2:], raw_ref, 2e-4, "raw fp32 output")
    close(_planes(act[:, 2:], co), F.elu(raw[:, 2:].cpu()), 4e-5, "ELU split planes")
    assert float(raw[:, :2].abs().max()) == 0.0 and float(_planes(act[:, :2], co).abs().max()) == 0.0
    # consumer: causal conv k=3 over the split planes (+ ELU split output), then k=1 conv with the raw skip operand
    hid = 32
    w1, b1 = rnd(hid, co, 3, seed=24, scale=(3 * co) ** -0.5), rnd(hid, seed=25)
    w2, b2 = rnd(co, hid, 1, seed=26, scale=hid ** -0.5), rnd(co, seed=27)
    h = raw[:, 2:].cpu()
    y1_ref = F.conv1d(F.pad(F.elu(h).transpose(1, 2), (2, 0)), w1, b1).transpose(1, 2)
    out_ref = F.elu(h + F.conv1d(F.elu(y1_ref).transpose(1, 2), w2, b2).transpose(1, 2))
    M2 = B * T * r
    y1 = torch.zeros(M2, hid, device=DEV)
    hip.gemm(act, hip.pack_w_bf16x3(dev(pack.pack_conv1d(w1))), y1, M=M2, N=hid, K=3 * co, lda=co, bias=dev(b1), rows_per_seg=T * r,
             a_seg_stride=(2 + T * r) * co, a_split=True, c_mode=1)
    close(_planes(y1, hid), F.elu(y1_ref).reshape(M2, hid), 2e-4, "conv k=3 on split planes")
    hip.gemm(y1, hip.pack_w_bf16x3(dev(pack.pack_conv1d(w2))), act, M=M2, N=co, K=hid, bias=dev(b2), epilogue=hip.EPI_RES, R=raw,
             rows_per_seg=T * r, a_split=True, c_off=2 * co, r_off=2 * co, c_seg_stride=(2 + T * r) * co, r_seg_stride=(2 + T * r) * co,
             ldc=co, ldr=co, c_mode=1)
    close(_planes(act[:, 2:], co), out_ref, 3e-4, "residual block output, ELU split")
    with pytest.raises(hip.SoproHipError):  # split operands are a property of the packed path
        hip.gemm(act, dev(pack.pack_conv1d(w1)), y1, M=M2, N=hid, K=3 * co, lda=co, a_split=True)


def test_gemm_bf16x3_activated_copy_outputs():
    """c_mode 3: ELU(C) as fp32 to C; c_mode 4: C to C and ELU(C) to C2 (SEANet decoder's producer-side activation)."""
    M, N, K = 300, 192, 128
    A, W, b, R = rnd(M, K, seed=31), rnd(N, K, seed=32, scale=K ** -0.5), rnd(N, seed=33), rnd(M, N, seed=34)
    Wp = hip.pack_w_bf16x3(dev(W))
    ref = A @ W.t() + b
    C, C2 = torch.empty(M, N, device=DEV), torch.empty(M, N, device=DEV)
    hip.gemm(dev(A), Wp, C, M=M, N=N, K=K, bias=dev(b), c_mode=3)
    close(C, F.elu(ref), 2e-4, "elu out")
    hip.gemm(dev(A), Wp, C, M=M, N=N, K=K, bias=dev(b), c_mode=4, C2=C2)
    close(C, ref, 2e-4, "raw out")
    close(C2, F.elu(ref), 2e-4, "elu copy")
    hip.gemm(dev(A), Wp, C, M=M, N=N, K=K, bias=dev(b), epilogue=hip.EPI_RES, R=dev(R), c_mode=3)
    close(C, F.elu(R + ref), 3e-4, "elu(res) out")


def test_mimi_decode_standalone_codec_matches_the_oracle(mc, mimi_np, mw):
    """A MimiCodec on its own (its own stage engine with the decoder family only): sopro_mimi_decode against the oracle
    decoder, a ragged-free batch of two, and the recorded call replayed."""
    from sopro_amd.codec import MimiCodec

    tok = torch.from_numpy(np.random.default_rng(3).integers(0, 2048, size=(2, 24, 32)))
    a = MimiCodec(mimi_np, mc, device=DEV)
    ya = a.decode_batch(tok)
    yb = a.decode_batch(tok)  # second call of a shape records
    yc = a.decode_batch(tok)  # later ones replay
    assert torch.equal(ya, yb) and torch.equal(ya, yc)
    for b in range(2):
        want = O.decode_full(tok[b], mw, mc).reshape(-1)
        assert float((ya[b].cpu() - want).abs().max()) < 1e-4 * float(want.abs().max())

# ------------------------------------------------------------------------------------------- skinny
@pytest.mark.parametrize("B", [1, 7, 16, 32, 40])
def test_skinny_norm_head(B):
    K, N = 384, 2049
    X, nw, W, b = rnd(B, K, seed=20), 1 + 0.1 * rnd(K, seed=21), rnd(N, K, seed=22, scale=K ** -0.5), rnd(N, seed=23)
    Y = torch.full((B, N), float("nan"), device=DEV)
    hip.skinny(dev(X), dev(W * nw[None, :]), Y, B=B, N=N, K=K, rms_norm=True, eps=1e-6, bias=dev(b))  # norm weight folded into W
    close(Y, O.rmsnorm(X, nw) @ W.t() + b, 5e-5, f"skinny head B={B}")


@pytest.mark.parametrize("B", [1, 16, 17, 32, 40])
@pytest.mark.parametrize("mt,nt", [(2, 1), (1, 2), (2, 2)])
def test_skinny_workgroup_shapes_are_bit_identical(B, mt, nt, w):
    """mt x nt (16-row groups x column tiles per workgroup) only changes who reads what: every output element sees the same
    arithmetic in the same order, so all shapes must reproduce the 1 x 1 results bit for bit - head (odd tile count, partial-sum
    input), FF1 (norm + GELU), FF2 (K-slices out) and the GLU / ring-buffer tail (partial-sum input, ring writes)."""
    D, K4, V1 = 384, 1536, 2049
    parts = dev(rnd(4, B, D, seed=910))
    pk = dict(Xp=parts[1:], np_=3, xp_stride=B * D)
    Wh, bh = dev(rnd(V1, D, seed=911, scale=D ** -0.5)), dev(rnd(V1, seed=912))
    W1, b1 = dev(rnd(K4, D, seed=913, scale=D ** -0.5)), dev(rnd(K4, seed=914))
    W2, b2, R = dev(rnd(D, K4, seed=915, scale=K4 ** -0.5)), dev(rnd(D, seed=916)), dev(rnd(B, D, seed=917))
    U = dev(rnd(B, K4, seed=918))
    p = "ar.blocks.1"
    k, dil = 13, 2
    L = (k - 1) * dil + 1
    gw, gb = dev(w[p + ".glu.pro.weight"] * w[p + ".norm.weight"][None, :]), dev(w[p + ".glu.pro.bias"])
    dww, dwb = dev(pack.pack_dw(w[p + ".dw.dw.weight"])), dev(w[p + ".dw.dw.bias"])
    ring0 = dev(rnd(L, B, D, seed=919))
    step = torch.full((1,), 5, dtype=torch.int32, device=DEV)

    def run(mt_, nt_, packed):
        t = dict(mt=mt_, nt=nt_)
        P = (lambda W_, glu=False: hip.pack_skinny_w(W_, glu=glu)) if packed else (lambda W_, glu=False: W_)
        out = {}
        Y = torch.full((B, V1), float("nan"), device=DEV)
        hip.skinny(parts[0], P(Wh), Y, B=B, N=V1, K=D, rms_norm=True, eps=1e-6, bias=bh, **pk, **t)
        out["head"] = Y
        Y = torch.full((B, K4), float("nan"), device=DEV)
        hip.skinny(parts[0], P(W1), Y, B=B, N=K4, K=D, rms_norm=True, eps=1e-6, bias=b1, epilogue=hip.EPI_GELU, **t)
        out["ff1"] = Y
        Y = torch.full((4, B, D), float("nan"), device=DEV)
        hip.skinny(U, P(W2), Y, B=B, N=D, K=K4, bias=b2, epilogue=hip.EPI_RES, R=R, ksplit=True, y_part_stride=B * D, **t)
        out["ff2"] = Y
        ring = ring0.clone()
        Y = torch.full((B, D), float("nan"), device=DEV)
        hip.skinny(parts[0], P(gw, True), Y, B=B, N=2 * D, K=D, rms_norm=True, eps=1e-6, bias=gb, epilogue=hip.EPI_GLU_DW, ring=ring,
                   dw_w=dww, dw_b=dwb, step=step, ring_len=L, ring_bcap=B, dil=dil, ksize=k, **pk, **t)
        out["glu"], out["ring"] = Y, ring
        torch.cuda.synchronize()
        return out

    for packed in (True, False):
        ref, got = run(1, 1, packed), run(mt, nt, packed)
        for name in ref:
            assert bool(torch.isfinite(ref[name]).all()), name
            assert torch.equal(ref[name], got[name]), (name, packed, mt, nt, float((ref[name] - got[name]).abs().max()))


def test_skinny_epilogues():
    B, K, N = 32, 1536, 384
    X, W, b, R, sc = rnd(B, K, seed=24), rnd(N, K, seed=25, scale=K ** -0.5), rnd(N, seed=26), rnd(B, N, seed=27), rnd(N, seed=28)
    ref = X @ W.t() + b
    Y = torch.empty(B, N, device=DEV)
    hip.skinny(dev(X), dev(W), Y, B=B, N=N, K=K, bias=dev(b), epilogue=hip.EPI_GELU)
    close(Y, F.gelu(ref), 5e-5, "gelu")
    Rd = dev(R)
    hip.skinny(dev(X), dev(W), Rd, B=B, N=N, K=K, bias=dev(b), epilogue=hip.EPI_RES, R=Rd, scale=dev(sc))
    close(Rd, R + sc * ref, 1e-4, "res in place")


def test_skinny_ksplit_partials_are_summed_by_the_consumer(w):
    """FF2 as 4 K-slices on 4x the workgroups (slice 0 carries bias + residual); the next kernel adds the slices in
    a fixed order while staging (deterministic) and can publish the combined stream (Xc)."""
    B, K, D = 32, 1536, 384
    U, W2, b2, R = rnd(B, K, seed=600), rnd(D, K, seed=601, scale=K ** -0.5), rnd(D, seed=602), rnd(B, D, seed=603)
    P = torch.full((4, B, D), float("nan"), device=DEV)
    hip.skinny(dev(U), dev(W2), P, B=B, N=D, K=K, bias=dev(b2), epilogue=hip.EPI_RES, R=dev(R), ksplit=True, y_part_stride=B * D)
    x2 = R + b2 + (U @ W2.t())
    close(P.sum(0), x2, 1e-4, "sum of K-slices (slice 0 holds bias + residual)")
    close(P[1:].sum(0) + P[0] - R.to(DEV) - b2.to(DEV), U @ W2.t(), 1e-4, "raw partial sums")
    pk = dict(Xp=P[1:], np_=3, xp_stride=B * D)
    nw, Wq = 1 + 0.1 * rnd(D, seed=604), rnd(D, D, seed=605, scale=D ** -0.5)
    Wqf = dev(Wq * nw[None, :])
    Y = torch.empty(B, D, device=DEV)
    hip.skinny(P[0], Wqf, Y, B=B, N=D, K=D, rms_norm=True, eps=1e-6, **pk)
    close(Y, O.rmsnorm(x2, nw) @ Wq.t(), 1e-4, "consumer of partials")
    # the GLU / ring-buffer tail adds its result to the combined input
    p = "ar.blocks.1"
    k, dil = 13, 2
    L = (k - 1) * dil + 1
    ring = torch.zeros(L, B, D, device=DEV)
    step = torch.zeros(1, dtype=torch.int32, device=DEV)
    gwf = dev(w[p + ".glu.pro.weight"] * w[p + ".norm.weight"][None, :])
    hip.skinny(P[0], gwf, Y, B=B, N=2 * D, K=D, rms_norm=True, eps=1e-6,
               bias=dev(w[p + ".glu.pro.bias"]), epilogue=hip.EPI_GLU_DW, ring=ring, dw_w=dev(pack.pack_dw(w[p + ".dw.dw.weight"])),
               dw_b=dev(w[p + ".dw.dw.bias"]), step=step, ring_len=L, ring_bcap=B, dil=dil, ksize=k, **pk)
    h = O.glu(O.rmsnorm(x2, w[p + ".norm.weight"]), w, p + ".glu")
    y = h * w[p + ".dw.dw.weight"][:, 0, -1] + w[p + ".dw.dw.bias"]  # empty ring: only the newest tap
    close(Y, x2 + y, 1e-4, "glu tail on a partial-sum input")
    # run-to-run bit reproducibility (no atomics anywhere on this path)
    Y2 = torch.empty(B, D, device=DEV)
    hip.skinny(P[0], Wqf, Y2, B=B, N=D, K=D, rms_norm=True, eps=1e-6, **pk)
    hip.skinny(P[0], Wqf, Y, B=B, N=D, K=D, rms_norm=True, eps=1e-6, **pk)
    assert torch.equal(Y, Y2)


@pytest.mark.parametrize("S,np_", [(19, 0), (64, 3), (130, 3)])
def test_xattn_step_folded_block_matches_text_xattn(S, np_, w):
    """One-launch cross-attention block on folded operands == TextXAttnBlock.forward with cached K/V
    (src/sopro/nn/text.py:85-132), including the partial-sum input and the per-head partial outputs."""
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
    Kp = torch.zeros(B, H, S_cap, D)
    Vp = torch.zeros(B, H, S_cap, D)
    for h in range(H):
        Kp[:, h, :S] = (k[:, h] @ Wq[h * dh:(h + 1) * dh]) * w[p + ".nq.weight"]   # [B,S,dh] @ [dh,D], RMSNorm weight folded in
        Vp[:, h, :S] = v[:, h] @ Wo[:, h * dh:(h + 1) * dh].t()       # [B,S,dh] @ [dh,D]
    Y = torch.full((H, B, D), float("nan"), device=DEV)
    Pd = dev(torch.stack(parts))
    hip.xattn_step(Pd[0], Y, None, dev(Kp), dev(Vp), dev(torch.tensor(klens, dtype=torch.int32)), B=B, H=H, D=D,
                   S_cap=S_cap, gate=float(torch.tanh(w[p + ".gate"])), scale=dh ** -0.5, eps=1e-6, Xp=Pd[1:] if np_ else None, np_=np_,
                   xp_stride=B * D, y_part_stride=B * D)
    close(Y.sum(0), ref, 5e-5, "folded cross-attention block")


@pytest.mark.parametrize("B,mt,nt", [(32, 1, 1), (64, 1, 2), (21, 2, 2)])
def test_skinny_aux_tiles_equal_separate_launches(B, mt, nt):
    """Round 4: aux column tiles (sopro_skinny_args.aux_*) - a second operand set on the rows a launch stages anyway - are the same
    function as a launch of their own, bit for bit, and leave the main tiles' results untouched: the FF1 form (main: RMSNorm ->
    projection -> GELU; aux: raw rows, no norm / bias / activation) and the FF2 form (K = 1536 as four K-slices with bias +
    residual in slice 0, for both operand sets)."""
    D = 384
    X, nw = rnd(B, D, seed=40), 1 + 0.1 * rnd(D, seed=41)
    W1, b1 = (rnd(4 * D, D, seed=42, scale=D ** -0.5) * nw[None, :]).contiguous(), rnd(4 * D, seed=43)
    Wa = rnd(D, D, seed=44, scale=D ** -0.5)
    kw = dict(mt=mt, nt=nt)
    W1p, Wap = hip.pack_skinny_w(dev(W1)), hip.pack_skinny_w(dev(Wa))
    Xd = dev(X)
    Y0, Ya0 = torch.full((B, 4 * D), float("nan"), device=DEV), torch.full((B, D), float("nan"), device=DEV)
    hip.skinny(Xd, W1p, Y0, B=B, N=4 * D, K=D, rms_norm=True, eps=1e-6, bias=dev(b1), epilogue=hip.EPI_GELU, **kw)
    hip.skinny(Xd, Wap, Ya0, B=B, N=D, K=D, **kw)
    Y1, Ya1 = torch.full((B, 4 * D), float("nan"), device=DEV), torch.full((B, D), float("nan"), device=DEV)
    hip.skinny(Xd, W1p, Y1, B=B, N=4 * D, K=D, rms_norm=True, eps=1e-6, bias=dev(b1), epilogue=hip.EPI_GELU, aux_W=Wap, aux_Y=Ya1, aux_flags=3, **kw)
    torch.cuda.synchronize()
    assert torch.equal(Y1, Y0) and torch.equal(Ya1, Ya0)
    close(Ya1, X @ Wa.t(), 5e-5, "aux projection of the raw rows")
    # FF2 form
    U, W2, b2, R = rnd(B, 4 * D, seed=45), rnd(D, 4 * D, seed=46, scale=(4 * D) ** -0.5), rnd(D, seed=47), rnd(B, D, seed=48)
    Wu, bu, Ru = rnd(D, 4 * D, seed=49, scale=(4 * D) ** -0.5), rnd(D, seed=50), rnd(B, D, seed=51)
    W2p, Wup, Ud = hip.pack_skinny_w(dev(W2)), hip.pack_skinny_w(dev(Wu)), dev(U)
    P0, Q0 = torch.full((4, B, D), float("nan"), device=DEV), torch.full((4, B, D), float("nan"), device=DEV)
    hip.skinny(Ud, W2p, P0, B=B, N=D, K=4 * D, bias=dev(b2), epilogue=hip.EPI_RES, R=dev(R), ksplit=True, y_part_stride=B * D, **kw)
    hip.skinny(Ud, Wup, Q0, B=B, N=D, K=4 * D, bias=dev(bu), epilogue=hip.EPI_RES, R=dev(Ru), ksplit=True, y_part_stride=B * D, **kw)
    P1, Q1 = torch.full((4, B, D), float("nan"), device=DEV), torch.full((4, B, D), float("nan"), device=DEV)
    hip.skinny(Ud, W2p, P1, B=B, N=D, K=4 * D, bias=dev(b2), epilogue=hip.EPI_RES, R=dev(R), ksplit=True, y_part_stride=B * D,
               aux_W=Wup, aux_Y=Q1, aux_bias=dev(bu), aux_R=dev(Ru), aux_y_part_stride=B * D, **kw)
    torch.cuda.synchronize()
    assert torch.equal(P1, P0) and torch.equal(Q1, Q0)
    close(Q1.sum(0), Ru + bu + U @ Wu.t(), 1e-4, "aux K-slices")


@pytest.mark.parametrize("S,np_", [(19, 0), (64, 3), (130, 3)])
def test_xattn_step_unfolded_keys_match_text_xattn(S, np_, w):
    """Round 4 (sopro_xattn_args.k_unfolded): the cross-attention block on UNFOLDED keys K [B, S_cap, D] with the raw query handed
    in as K-slice partials == TextXAttnBlock.forward with cached K/V (src/sopro/nn/text.py:85-132), and == the folded-key kernel
    to round-off (same softmax, same folded V')."""
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
    Kp, Vp, Ku = torch.zeros(B, H, S_cap, D), torch.zeros(B, H, S_cap, D), torch.zeros(B, S_cap, D)
    for h in range(H):
        Kp[:, h, :S] = (k[:, h] @ Wq[h * dh:(h + 1) * dh]) * w[p + ".nq.weight"]
        Vp[:, h, :S] = v[:, h] @ Wo[:, h * dh:(h + 1) * dh].t()
        Ku[:, :S, h * dh:(h + 1) * dh] = k[:, h]
    q_raw = (x.double() @ (Wq.double() * w[p + ".nq.weight"].double()[None, :]).t()).float()  # Wq' x (the frame gets it from FF1 / FF2 aux tiles)
    g = torch.Generator().manual_seed(3)
    cuts = torch.rand(3, B, D, generator=g)
    qparts = torch.stack([q_raw * cuts[0], q_raw * (1 - cuts[0]) * cuts[1], q_raw * (1 - cuts[0]) * (1 - cuts[1]) * cuts[2],
                          q_raw * (1 - cuts[0]) * (1 - cuts[1]) * (1 - cuts[2])])
    Pd = dev(torch.stack(parts))
    kl = dev(torch.tensor(klens, dtype=torch.int32))
    kw = dict(B=B, H=H, D=D, S_cap=S_cap, gate=float(torch.tanh(w[p + ".gate"])), scale=dh ** -0.5, eps=1e-6, Xp=Pd[1:] if np_ else None, np_=np_,
              xp_stride=B * D, y_part_stride=B * D)
    Yf, Yu = torch.full((H, B, D), float("nan"), device=DEV), torch.full((H, B, D), float("nan"), device=DEV)
    hip.xattn_step(Pd[0], Yf, None, dev(Kp), dev(Vp), kl, **kw)
    hip.xattn_step(Pd[0], Yu, None, dev(Ku), dev(Vp), kl, Qp=dev(qparts), nqp=4, qp_stride=B * D, **kw)
    close(Yu.sum(0), ref, 5e-5, "unfolded keys vs TextXAttnBlock")
    close(Yu.sum(0), Yf.sum(0), 2e-5, "unfolded vs folded keys")
    # bf16 mode: the same with bf16 K / V'
    K16, V16 = dev(Ku).to(torch.bfloat16), dev(Vp).to(torch.bfloat16)
    Y16, Y32 = torch.full((H, B, D), float("nan"), device=DEV), torch.full((H, B, D), float("nan"), device=DEV)
    hip.xattn_step(Pd[0], Y16, None, K16, V16, kl, Qp=dev(qparts), nqp=4, qp_stride=B * D, **kw)
    hip.xattn_step(Pd[0], Y32, None, K16.float(), V16.float(), kl, Qp=dev(qparts), nqp=4, qp_stride=B * D, **kw)
    close(Y16.sum(0), Y32.sum(0), 2e-5, "bf16-stored unfolded keys vs the same values in fp32")


@pytest.mark.parametrize("S", [5, 64, 130])
def test_attention_decode_single_query(S):
    B, H, dh = 5, 4, 96
    D = H * dh
    q, kv = rnd(B, D, seed=610), rnd(B, S, 2 * D, seed=611)
    klens = [S, 1, max(1, S // 2), S, max(1, S - 3)]
    keep = torch.arange(S)[None, :] < torch.tensor(klens)[:, None]
    ref = O._unheads(O.attention(O._heads(q[:, None], H), O._heads(kv[..., :D].contiguous(), H), O._heads(kv[..., D:].contiguous(), H), keep))[:, 0]
    out = torch.full((B, D), float("nan"), device=DEV)
    kvd = dev(kv)
    hip.attention(dev(q), kvd, kvd, out, B=B, H=H, dh=dh, Tq=1, Tk=S, ldq=D, ldk=2 * D, ldv=2 * D, ldo=D, q_bstride=D,
                  k_bstride=S * 2 * D, v_bstride=S * 2 * D, o_bstride=D, klens=dev(torch.tensor(klens, dtype=torch.int32)), v_off=D,
                  decode=True)
    close(out, ref, 2e-5, "decode attention")


@pytest.mark.parametrize("B,dil", [(1, 1), (5, 2), (32, 4)])
def test_skinny_glu_ring_buffer_step_matches_forward_step(B, dil, w, cfg):
    """SSMLiteBlock.forward_step first half over 30 frames (src/sopro/nn/blocks.py:150-157, 76-110)."""
    D, k = 384, 13
    p = "ar.blocks.2"
    L = (k - 1) * dil + 1
    ring_o = torch.zeros(B, L, D)
    ring = torch.zeros(L, B, D, device=DEV)
    step = torch.zeros(1, dtype=torch.int32, device=DEV)
    gw, gb = dev(w[p + ".glu.pro.weight"] * w[p + ".norm.weight"][None, :]), dev(w[p + ".glu.pro.bias"])
    dww, dwb = dev(pack.pack_dw(w[p + ".dw.dw.weight"])), dev(w[p + ".dw.dw.bias"])
    Y = torch.empty(B, D, device=DEV)
    for t in range(30):
        x = rnd(B, D, seed=100 + t)
        h = O.glu(O.rmsnorm(x, w[p + ".norm.weight"]), w, p + ".glu")
        ring_o = torch.cat([ring_o[:, 1:], h.unsqueeze(1)], dim=1)
        taps = ring_o[:, torch.arange(0, k * dil, dil)]
        y = (taps.transpose(1, 2) * w[p + ".dw.dw.weight"].squeeze(1)).sum(-1) + w[p + ".dw.dw.bias"]
        step.fill_(t)
        hip.skinny(dev(x), gw, Y, B=B, N=2 * D, K=D, rms_norm=True, eps=1e-6, bias=gb, epilogue=hip.EPI_GLU_DW, ring=ring,
                   dw_w=dww, dw_b=dwb, step=step, ring_len=L, ring_bcap=B, dil=dil, ksize=k)
        close(Y, x + y, 5e-5, f"glu+dw step {t}")


# ------------------------------------------------------------------------------- norms / elementwise
def test_norms_and_small_ops():
    rows, C = 77, 384
    x, wt, b = rnd(rows, C, seed=30), 1 + 0.1 * rnd(C, seed=31), rnd(C, seed=32)
    out = torch.empty(rows, C, device=DEV)
    hip.norm(dev(x), out, dev(wt), rows=rows, C_=C, eps=1e-6)
    close(out, O.rmsnorm(x, wt), 1e-5, "rmsnorm")
    x5 = rnd(rows, 512, seed=33)
    o5 = torch.empty(rows, 512, device=DEV)
    w5, b5 = 1 + 0.1 * rnd(512, seed=34), rnd(512, seed=35)
    hip.norm(dev(x5), o5, dev(w5), rows=rows, C_=512, eps=1e-5, kind=hip.NORM_LN, b=dev(b5))
    close(o5, F.layer_norm(x5, (512,), w5, b5, 1e-5), 2e-5, "layernorm")
    # FiLM-style per-segment modulation: 7 segments of 11 rows
    mul, add = rnd(7, C, seed=36), rnd(7, C, seed=37)
    hip.norm(dev(x), out, dev(wt), rows=rows, C_=C, eps=1e-5, kind=hip.NORM_LN, b=dev(b), mul=dev(mul), add=dev(add), rows_per_seg=11)
    ref = F.layer_norm(x, (C,), wt, b, 1e-5).view(7, 11, C) * mul[:, None] + add[:, None]
    close(out, ref.view(rows, C), 3e-5, "layernorm + film")
    # segmented source rows (zero-padded stream)
    buf = torch.zeros(7, 3 + 11, C)
    buf[:, 3:] = x.view(7, 11, C)
    hip.norm(dev(buf), out, dev(wt), rows=rows, C_=C, eps=1e-6, rows_per_seg=11, x_off=3 * C, x_seg_stride=14 * C)
    close(out, O.rmsnorm(x, wt), 1e-5, "rmsnorm segmented")
    a = rnd(rows, C, seed=38, scale=3.0)
    hip.rms_match(dev(a), dev(x), out, rows, C)
    rms = lambda t: torch.sqrt(t.pow(2).mean(-1, keepdim=True) + 1e-6)  # noqa: E731
    close(out, a * (rms(x) / rms(a)).clamp(0, 10), 1e-5, "rms_match")
    hip.tanh_affine(dev(x), out, 1.0, 1.2, rows * C)
    close(out, 1.0 + 1.2 * torch.tanh(x), 1e-6, "tanh_affine")
    pe = pack.sinusoid_table(64, C)
    o3 = torch.empty(3, 20, C, device=DEV)
    hip.add_pos(dev(x[:3]), dev(pe), o3, 3, 20, C, 5)
    close(o3, x[:3, None] + pe[None, 5:25], 1e-6, "add_pos")
    lens = torch.tensor([11, 4, 9], dtype=torch.int32)
    xm = rnd(3, 11, C, seed=39)
    om = torch.empty(3, C, device=DEV)
    hip.masked_mean(dev(xm), dev(lens), om, 3, 11, C)
    ref = torch.stack([xm[i, : lens[i]].sum(0) / (float(lens[i]) + 1e-6) for i in range(3)])
    close(om, ref, 1e-5, "masked_mean")
    lg = rnd(3, 11, seed=40)
    st = torch.empty(3, 2 * C, device=DEV)
    hip.stats_pool(dev(xm), dev(lg), None, st, 3, 11, C)
    aw = torch.softmax(lg, 1).unsqueeze(-1)
    mu = (xm * aw).sum(1)
    sd = torch.sqrt((aw * (xm - mu[:, None]).pow(2)).sum(1).clamp_min(1e-6))
    close(st, torch.cat([mu, sd], -1), 1e-5, "stats_pool")
    e = rnd(4, 192, seed=41)
    oe = torch.empty(4, 192, device=DEV)
    hip.l2norm(dev(e), oe, 4, 192, 1e-6)
    close(oe, F.normalize(e, dim=-1, eps=1e-6), 1e-6, "l2norm")
    am = torch.empty(rows, dtype=torch.int32, device=DEV)
    xx = rnd(rows, 2048, seed=42)
    hip.argmax_rows(dev(xx), am, rows=rows, N=2048)
    assert torch.equal(am.cpu().long(), xx.argmax(-1))


@pytest.mark.parametrize("ksize,dil,causal", [(7, 1, False), (11, 8, False), (11, 2, False), (13, 4, True)])
def test_dwconv(ksize, dil, causal):
    B, T, C = 3, 50, 384
    x, wt, b, res = rnd(B, T, C, seed=50), rnd(C, 1, ksize, seed=51), rnd(C, seed=52), rnd(B, T, C, seed=53)
    total = (ksize - 1) * dil
    left = total if causal else total // 2
    ref = O.dwconv_full(x, wt, b, dil, causal)
    out = torch.empty(B, T, C, device=DEV)
    hip.dwconv(dev(x), dev(pack.pack_dw(wt)), dev(b), out, B=B, T=T, C_=C, ksize=ksize, dil=dil, left=left, mode=1, res=dev(res))
    close(out, res + ref, 2e-5, "dwconv + res")
    hip.dwconv(dev(x), dev(pack.pack_dw(wt)), dev(b), out, B=B, T=T, C_=C, ksize=ksize, dil=dil, left=left, mode=2)
    close(out, F.gelu(ref), 2e-5, "dwconv + gelu")
    # ragged: utterance b is zero-padded at its own end, like a single-utterance call of the reference
    lens = [50, 17, 33]
    hip.dwconv(dev(x), dev(pack.pack_dw(wt)), dev(b), out, B=B, T=T, C_=C, ksize=ksize, dil=dil, left=left, mode=0,
               lens=dev(torch.tensor(lens, dtype=torch.int32)))
    for i, n in enumerate(lens):
        close(out[i, :n], O.dwconv_full(x[i: i + 1, :n], wt, b, dil, causal)[0], 2e-5, f"dwconv ragged {i}")


@pytest.mark.parametrize("ksize,dil,causal", [(11, 1, False), (11, 2, False), (11, 4, False), (11, 8, False), (7, 1, False), (7, 3, True)])
def test_dwconv_comb_form_is_the_same_function(ksize, dil, causal):
    """Round 5: long inputs run the comb form of the depthwise convolution (a thread = 8 outputs of one residue class mod the
    dilation: every input row is loaded once per 8 outputs instead of once per tap).  Same taps in the same order per output:
    bit-identical to the per-output kernel - which single utterances still take (fewer than 1024 rows per call) - with ragged
    lengths, every epilogue mode, T not a multiple of 8 x dilation; and equal to the oracle's convolution."""
    B, T, C = 7, 203, 384
    x, wt, b, res = rnd(B, T, C, seed=54), rnd(C, 1, ksize, seed=55), rnd(C, seed=56), rnd(B, T, C, seed=57)
    total = (ksize - 1) * dil
    left = total if causal else total // 2
    wd, bd, xd, rd = dev(pack.pack_dw(wt)), dev(b), dev(x), dev(res)
    lens = [203, 17, 200, 96, 1, 150, 203]
    ld = dev(torch.tensor(lens, dtype=torch.int32))
    for mode in (0, 1, 2):
        for use_lens in (False, True):
            kw = dict(C_=C, ksize=ksize, dil=dil, left=left, mode=mode, res=rd if mode == 1 else None)
            comb = torch.full((B, T, C), float("nan"), device=DEV)
            hip.dwconv(xd, wd, bd, comb, B=B, T=T, lens=ld if use_lens else None, **kw)  # 1421 rows: the comb form
            for i in range(B):  # one utterance per call: the per-output kernel
                one = torch.full((1, T, C), float("nan"), device=DEV)
                hip.dwconv(xd[i: i + 1].contiguous(), wd, bd, one, B=1, T=T, lens=ld[i: i + 1].contiguous() if use_lens else None,
                           **dict(kw, res=rd[i: i + 1].contiguous() if mode == 1 else None))
                assert torch.equal(comb[i], one[0]), (mode, use_lens, i)
    out = torch.empty(B, T, C, device=DEV)
    hip.dwconv(xd, wd, bd, out, B=B, T=T, C_=C, ksize=ksize, dil=dil, left=left, mode=0, lens=ld)
    for i, n in enumerate(lens):
        close(out[i, :n], O.dwconv_full(x[i: i + 1, :n], wt, b, dil, causal)[0], 2e-5, f"comb dwconv ragged {i}")


def test_gathers():
    V, Q, D, rows = 2048, 32, 384, 41
    table = rnd(Q * V + 1, D, seed=60)
    tok = torch.randint(0, V, (rows, Q), generator=torch.Generator().manual_seed(61), dtype=torch.int32)
    cols = [0, 3, 4, 9]
    wq = torch.softmax(rnd(len(cols), seed=62), 0)
    base = rnd(rows, D, seed=63)
    out = torch.empty(rows, D, device=DEV)
    i32 = lambda v: torch.tensor(v, dtype=torch.int32, device=DEV)  # noqa: E731
    hip.codebook_sum(dev(tok), Q, i32(cols), i32([c * V for c in cols]), dev(wq), dev(table), out, rows=rows, D=D, base=dev(base),
                     alpha=0.3, beta=0.7)
    ref = 0.3 * base + 0.7 * sum(wq[j] * table[c * V + tok[:, c].long()] for j, c in enumerate(cols))
    close(out, ref, 1e-5, "codebook_sum")
    ids = torch.randint(0, 500, (2, 9), generator=torch.Generator().manual_seed(64), dtype=torch.int32)
    lens = torch.tensor([9, 5], dtype=torch.int32)
    tt, pe = rnd(500, D, seed=65), pack.sinusoid_table(32, D)
    o2 = torch.empty(2, 9, D, device=DEV)
    hip.text_embed(dev(ids), dev(lens), dev(tt), dev(pe), o2, 2, 9, D)
    ref = tt[ids.long()] + pe[None, :9]
    ref[1, 5:] = 0
    close(o2, ref, 1e-6, "text_embed")


# ---------------------------------------------------------------------------------------- attention
@pytest.mark.parametrize("H,dh,Tq,Tk", [(4, 96, 1, 19), (4, 96, 1, 130), (2, 192, 49, 30), (2, 192, 401, 150), (8, 64, 40, 40), (4, 96, 33, 70), (2, 192, 201, 150), (2, 192, 16, 2)])
def test_attention_cross(H, dh, Tq, Tk):
    B, D = 3, H * dh
    q, k, v = rnd(B, Tq, D, seed=70), rnd(B, Tk, D, seed=71), rnd(B, Tk, D, seed=72)
    klens = [Tk, max(1, Tk // 3), Tk - 1]
    keep = torch.arange(Tk)[None, :] < torch.tensor(klens)[:, None]
    ref = O._unheads(O.attention(O._heads(q, H), O._heads(k, H), O._heads(v, H), keep))
    out = torch.empty(B, Tq, D, device=DEV)
    hip.attention(dev(q), dev(k), dev(v), out, B=B, H=H, dh=dh, Tq=Tq, Tk=Tk, ldq=D, ldk=D, ldv=D, ldo=D, q_bstride=Tq * D,
                  k_bstride=Tk * D, v_bstride=Tk * D, o_bstride=Tq * D, klens=dev(torch.tensor(klens, dtype=torch.int32)))
    close(out, ref, 2e-5, "cross attention")


@pytest.mark.parametrize("N,win,past", [(100, 250, 0), (400, 250, 0), (16, 250, 300), (300, 17, 5), (33, 250, 0), (7, 250, 40), (95, 40, 1000)])
def test_attention_causal_sliding_window_with_cache(N, win, past):
    """Mimi decoder attention incl. the cached-keys form (HF:modeling_mimi.py:657-726, 882-888)."""
    B, H, dh = (2 if past == 0 else 1), 8, 64
    D = H * dh
    Tk = N + min(past, win - 1)
    q, kv = rnd(B, N, D, seed=73), rnd(B, Tk, 2 * D, seed=74)
    pos = torch.arange(N) + past
    kpos = torch.arange(past + N - Tk, past + N)
    vis = (kpos[None, :] <= pos[:, None]) & (kpos[None, :] > pos[:, None] - win)
    s = torch.matmul(O._heads(q, H), O._heads(kv[..., :D].contiguous(), H).transpose(-1, -2)) / math.sqrt(dh)
    s = s.masked_fill(~vis[None, None], float("-inf"))
    ref = O._unheads(torch.matmul(torch.softmax(s, -1), O._heads(kv[..., D:].contiguous(), H)))
    out = torch.empty(B, N, D, device=DEV)
    kvd = dev(kv)
    hip.attention(dev(q), kvd, kvd, out, B=B, H=H, dh=dh, Tq=N, Tk=Tk, ldq=D, ldk=2 * D, ldv=2 * D, ldo=D, q_bstride=N * D,
                  k_bstride=Tk * 2 * D, v_bstride=Tk * 2 * D, o_bstride=N * D, causal=True, window=win, q_pos0=past,
                  k_pos0=past + N - Tk, v_off=D)
    close(out, ref, 2e-5, "causal window attention")


@pytest.mark.parametrize("N,win,past,passes", [(400, 250, 0, 3), (100, 250, 0, 3), (16, 250, 300, 3), (300, 17, 5, 3), (33, 250, 0, 3), (95, 40, 1000, 3),
                                               (400, 250, 0, 1), (7, 250, 40, 3)])
def test_attention_window_split_bf16_form(N, win, past, passes):
    """The decoder's waveform-path form of the window attention (two bf16 pieces / three MFMA passes, or one piece): same
    masks and cache geometry as the exact kernel, error at the level of 16-bit (8-bit) operands.  < 16 queries: exact kernel."""
    B, H, dh = (2 if past == 0 else 1), 8, 64
    D = H * dh
    Tk = N + min(past, win - 1)
    q, kv = rnd(B, N, D, seed=75), rnd(B, Tk, 2 * D, seed=76)
    pos = torch.arange(N) + past
    kpos = torch.arange(past + N - Tk, past + N)
    vis = (kpos[None, :] <= pos[:, None]) & (kpos[None, :] > pos[:, None] - win)
    s = torch.matmul(O._heads(q.double(), H), O._heads(kv[..., :D].contiguous().double(), H).transpose(-1, -2)) / math.sqrt(dh)
    s = s.masked_fill(~vis[None, None], float("-inf"))
    ref = O._unheads(torch.matmul(torch.softmax(s, -1), O._heads(kv[..., D:].contiguous().double(), H))).float()
    kvd = dev(kv)
    kw = dict(B=B, H=H, dh=dh, Tq=N, Tk=Tk, ldq=D, ldk=2 * D, ldv=2 * D, ldo=D, q_bstride=N * D, k_bstride=Tk * 2 * D,
              v_bstride=Tk * 2 * D, o_bstride=N * D, causal=True, window=win, q_pos0=past, k_pos0=past + N - Tk, v_off=D)
    out, exact = torch.empty(B, N, D, device=DEV), torch.empty(B, N, D, device=DEV)
    hip.attention(dev(q), kvd, kvd, out, split_passes=passes, **kw)
    hip.attention(dev(q), kvd, kvd, exact, **kw)
    close(exact, ref, 2e-5, "exact window attention")
    scale = float(ref.abs().max())
    err = float((out.cpu() - ref).abs().max()) / scale
    assert torch.isfinite(out).all()
    assert err < (3e-5 if passes == 3 else 2e-2), f"split window attention ({passes} passes): {err:.2e} of the largest output"
    if N < 16:
        assert torch.equal(out, exact)


def test_rope_upsample_final_conv():
    H, dh, rows_per_seg, B = 8, 64, 21, 2
    x = rnd(B * rows_per_seg, 3 * H * dh, seed=80)
    c, s = pack.rope_tables(64, dh, 10000.0)
    xd = dev(x)
    hip.rope(xd, dev(c), dev(s), rows=B * rows_per_seg, rows_per_seg=rows_per_seg, pos0=7, H=H, dh=dh, ldx=3 * H * dh, x_off=H * dh)
    cos, sin = O.rope_cos_sin(torch.arange(rows_per_seg) + 7, dh, 10000.0)
    kk = O._heads(x[:, H * dh: 2 * H * dh].reshape(B, rows_per_seg, H * dh), H)
    ref = O._unheads(kk * cos + O._rot_half(kk) * sin).reshape(B * rows_per_seg, H * dh)
    close(xd[:, H * dh: 2 * H * dh], ref, 1e-5, "rope")
    close(xd[:, : H * dh], x[:, : H * dh], 0.0, "rope leaves q alone")
    # upsample (HF:modeling_mimi.py:1208-1216)
    T, C = 9, 512
    xu, wu = rnd(B, T, C, seed=81), rnd(C, 1, 4, seed=82)
    ref = O.causal_convtr1d(xu.transpose(1, 2), wu, None, 2, groups=C).transpose(1, 2)
    y = torch.zeros(B, 6 + 2 * T, C, device=DEV)
    hip.upsample2(dev(xu), dev(wu.squeeze(1)), y, B=B, T=T, C_=C, y_seg_stride=(6 + 2 * T) * C, y_off=6 * C)
    close(y[:, 6:], ref, 1e-5, "upsample")
    # last conv (HF:modeling_mimi.py:957-960)
    Tn = 1000
    h, wf, bf = rnd(B, Tn, 64, seed=83), rnd(1, 64, 3, seed=84, scale=0.1), 0.05
    ref = O.causal_conv1d(F.elu(h).transpose(1, 2), wf, torch.tensor([bf]))[:, 0]
    hb = torch.zeros(B, 2 + Tn, 64)
    hb[:, 2:] = h
    wav = torch.empty(B, Tn, device=DEV)
    hip.final_conv(dev(hb), dev(wf[0].t()), bf, wav, B=B, T=Tn, h_seg_stride=(2 + Tn) * 64, wav_seg_stride=Tn)
    close(wav, ref, 2e-5, "final conv")


def test_seanet_tail_fused_matches_unfused_layers():
    """Last MimiResnetBlock + last conv in one kernel (HF:modeling_mimi.py:408-447, 957-960)."""
    B, T = 2, 300  # 3 tiles of 126 samples, the last one partial
    h = rnd(B, T, 64, seed=700)
    w1, b1 = rnd(32, 64, 3, seed=701, scale=0.07), rnd(32, seed=702, scale=0.1)
    w2, b2 = rnd(64, 32, 1, seed=703, scale=0.17), rnd(64, seed=704, scale=0.1)
    wf, bf = rnd(1, 64, 3, seed=705, scale=0.07), 0.03
    x = h.transpose(1, 2)
    y = O.causal_conv1d(F.elu(x), w1, b1)
    y = O.causal_conv1d(F.elu(y), w2, b2)
    ref = O.causal_conv1d(F.elu(x + y), wf, torch.tensor([bf]))[:, 0]
    hb = torch.zeros(B, 2 + T, 64)
    hb[:, 2:] = h
    wav = torch.full((B, T), float("nan"), device=DEV)
    hip.seanet_tail(dev(hb), dev(pack.pack_conv1d(w1)), dev(b1), dev(pack.pack_conv1d(w2)), dev(b2), dev(wf[0].t()), bf, wav,
                    B=B, T=T, h_seg_stride=(2 + T) * 64, wav_seg_stride=T)
    close(wav, ref, 1e-4, "fused SEANet tail")  # split-bf16 contractions (16 mantissa bits per operand)


def test_seanet_tail_tiles_per_workgroup_is_the_same_function():
    """Large inputs give a workgroup several consecutive tiles (weight fragments prepared once): bit-identical samples."""
    B, T = 3, 1000  # 8 tiles of 126 samples, the last one partial
    hb = torch.zeros(B, 2 + T, 64)
    hb[:, 2:] = rnd(B, T, 64, seed=710)
    w1, b1 = rnd(32, 192, seed=711, scale=0.07), rnd(32, seed=712, scale=0.1)
    w2, b2 = rnd(64, 32, seed=713, scale=0.17), rnd(64, seed=714, scale=0.1)
    wf = rnd(3, 64, seed=715, scale=0.07)
    args = [dev(x) for x in (hb, w1, b1, w2, b2, wf)]
    lib, outs = hip.load(), []
    try:
        for tiles in (1, 3, 8, 16):
            lib.sopro_seanet_tail_set_tiles(tiles)
            wav = torch.full((B, T), float("nan"), device=DEV)
            hip.seanet_tail(*args, 0.03, wav, B=B, T=T, h_seg_stride=(2 + T) * 64, wav_seg_stride=T)
            torch.cuda.synchronize()
            outs.append(wav.cpu())
    finally:
        lib.sopro_seanet_tail_set_tiles(0)
    assert bool(torch.isfinite(outs[0]).all())
    for o in outs[1:]:
        assert torch.equal(o, outs[0])


@pytest.mark.parametrize("B,T", [(2, 300), (3, 1000), (1, 62), (2, 8000)])
def test_seanet_tail_sixteen_wave_kernel(B, T):
    """The long-input form of the tail (four tile groups of 62 samples per workgroup over LDS-resident weight fragments,
    v_mfma_f32_16x16x32_bf16): against torch's convolutions and against the four-wave kernel (same operand rounding, another
    accumulation order)."""
    h = rnd(B, T, 64, seed=730)
    w1, b1 = rnd(32, 64, 3, seed=731, scale=0.07), rnd(32, seed=732, scale=0.1)
    w2, b2 = rnd(64, 32, 1, seed=733, scale=0.17), rnd(64, seed=734, scale=0.1)
    wf, bf = rnd(1, 64, 3, seed=735, scale=0.07), 0.03
    x = h.transpose(1, 2)
    y = O.causal_conv1d(F.elu(x), w1, b1)
    y = O.causal_conv1d(F.elu(y), w2, b2)
    ref = O.causal_conv1d(F.elu(x + y), wf, torch.tensor([bf]))[:, 0]
    hb = torch.zeros(B, 2 + T, 64)
    hb[:, 2:] = h
    args = [dev(t) for t in (hb, pack.pack_conv1d(w1), b1, pack.pack_conv1d(w2), b2, wf[0].t())]
    lib, outs = hip.load(), {}
    try:
        for tiles in (-1, 1):
            lib.sopro_seanet_tail_set_tiles(tiles)
            wav = torch.full((B, T), float("nan"), device=DEV)
            hip.seanet_tail(*args, bf, wav, B=B, T=T, h_seg_stride=(2 + T) * 64, wav_seg_stride=T)
            torch.cuda.synchronize()
            outs[tiles] = wav.cpu()
    finally:
        lib.sopro_seanet_tail_set_tiles(0)
    close(outs[-1], ref, 1e-4, "sixteen-wave SEANet tail")
    assert float((outs[-1] - outs[1]).abs().max()) < 2e-6 * float(ref.abs().max() + 1.0)


def test_seanet_res128_fused_block_matches_the_layers():
    """MimiResnetBlock(dim 128) + the next layer's ELU in one weight-stationary kernel (HF:modeling_mimi.py:408-447):
    against torch's convolutions, for one and several tiles per workgroup (bit-identical), partial last tile, batch of 3."""
    B, T = 3, 333  # 6 tiles of 64 rows, the last one partial
    h = rnd(B, T, 128, seed=720)
    w1, b1 = rnd(64, 128, 3, seed=721, scale=0.05), rnd(64, seed=722, scale=0.1)
    w2, b2 = rnd(128, 64, 1, seed=723, scale=0.12), rnd(128, seed=724, scale=0.1)
    x = h.transpose(1, 2)
    y = O.causal_conv1d(F.elu(x), w1, b1)
    y = O.causal_conv1d(F.elu(y), w2, b2)
    ref = F.elu(x + y).transpose(1, 2)
    hb = torch.zeros(B, 2 + T + 5, 128)  # segment stride larger than the rows in use
    hb[:, 2:2 + T] = h
    args = [dev(t) for t in (hb, pack.pack_conv1d(w1), b1, pack.pack_conv1d(w2), b2)]
    lib, outs = hip.load(), []
    try:
        for tiles in (0, 1, 2, 4):
            lib.sopro_seanet_res_set_tiles(tiles)
            out = torch.full((B, 2 + T + 5, 128), float("nan"), device=DEV)
            out[:, :2] = 0.0
            hip.seanet_res128(*args, out, B=B, T=T, h_seg_stride=(2 + T + 5) * 128, out_seg_stride=(2 + T + 5) * 128)
            torch.cuda.synchronize()
            outs.append(out.cpu())
    finally:
        lib.sopro_seanet_res_set_tiles(0)
    close(outs[0][:, 2:2 + T], ref, 1e-4, "fused 128-channel residual block")  # split-bf16 contractions (16 mantissa bits per operand)
    assert bool(torch.isnan(outs[0][:, 2 + T:]).all()) and float(outs[0][:, :2].abs().max()) == 0.0  # nothing outside rows 2 .. 2+T
    for o in outs[1:]:
        assert torch.equal(o[:, 2:2 + T], outs[0][:, 2:2 + T])
    with pytest.raises(hip.SoproHipError):
        hip.seanet_res128(args[0], *args[1:], args[0], B=B, T=T, h_seg_stride=(2 + T + 5) * 128, out_seg_stride=(2 + T + 5) * 128)


def test_host_mirror_copies_behind_the_queued_launches():
    """hip.HostMirror: page-locked int32 words outside torch's pinned-memory cache; the copy is ordered on the current stream."""
    m = hip.HostMirror(6)
    assert m.values() == [0] * 6
    t = torch.arange(6, dtype=torch.int32, device=DEV)
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        t2 = t * 7  # a launch the copy has to wait for
        m.copy_from(t2)
        ev = torch.cuda.Event()
        ev.record(s)
    ev.synchronize()
    assert m.values() == [0, 7, 14, 21, 28, 35]
    with pytest.raises(hip.SoproHipError):
        m.copy_from(t[:5])
    with pytest.raises(hip.SoproHipError):
        m.copy_from(t.to(torch.int64))  # (32-bit words only: int32 or float32)
    # the other direction: a kernel reads the page-locked words (parameter blocks of the stage sequences, round 5)
    m.array()[:] = [5, 4, 3, 2, 1, 0]
    back = torch.zeros(6, dtype=torch.int32, device=DEV)
    m.copy_to(back)
    torch.cuda.synchronize()
    assert back.tolist() == [5, 4, 3, 2, 1, 0]


@pytest.mark.parametrize("passes", [3, 1])
def test_seanet_up128_weight_stationary_equals_the_tile_kernel(passes):
    """Last transposed convolution of SEANet (ConvTranspose1d 128 -> 64, k = 8, s = 4; HF:modeling_mimi.py:931-961) in its
    weight-stationary form: against torch's conv_transpose1d, bit for bit against the generic split-bf16 tile kernel on the
    same operands (three-pass and bf16-mode one-pass), for any number of tiles per workgroup, partial last tile, batch of 3,
    segment strides larger than the rows in use."""
    B, T, ci, co, r = 3, 333, 128, 64, 4
    x = rnd(B, T, ci, seed=760)
    wt, bt = rnd(ci, co, 2 * r, seed=761, scale=0.06), rnd(co, seed=762, scale=0.1)
    ref = F.conv_transpose1d(x.transpose(1, 2), wt, bt, stride=r)[..., :T * r].transpose(1, 2)  # causal: trim the right tail
    W, bias = pack.pack_convtr1d(wt, bt, r)  # [r*co, 2*ci], [r*co]
    xs, os_ = (1 + T + 3) * ci, (2 + T * r + 7) * co
    xb = torch.zeros(B, 1 + T + 3, ci)
    xb[:, 1:1 + T] = x
    xd, Wd, bd = dev(xb), dev(W), dev(bias)
    lib, outs = hip.load(), []
    try:
        for tiles in (0, 1, 2, 5):
            lib.sopro_seanet_up_set_tiles(tiles)
            out = torch.full((B, 2 + T * r + 7, co), float("nan"), device=DEV)
            hip.seanet_up128(xd, Wd, bd, out, B=B, T=T, x_seg_stride=xs, out_seg_stride=os_, out_off=2 * co, passes=passes)
            torch.cuda.synchronize()
            outs.append(out.cpu())
    finally:
        lib.sopro_seanet_up_set_tiles(0)
    got = outs[0][:, 2:2 + T * r]
    close(got, ref, 1e-4 if passes == 3 else 2e-2, "weight-stationary transposed convolution")
    assert bool(torch.isnan(outs[0][:, :2]).all()) and bool(torch.isnan(outs[0][:, 2 + T * r:]).all())  # nothing outside its rows
    for o in outs[1:]:
        assert torch.equal(o[:, 2:2 + T * r], got)
    # the generic kernel on the same shape: row t of A = [x[t-1] | x[t]] (overlapping rows, lda = ci)
    Wp = hip.pack_w_bf16x3(Wd) if passes == 3 else hip.pack_w_bf16x1(Wd)
    out2 = torch.full((B, 2 + T * r + 7, co), float("nan"), device=DEV)
    hip.gemm(xd, Wp, out2, M=B * T, N=r * co, K=2 * ci, lda=ci, bias=bd, rows_per_seg=T, a_seg_stride=xs, c_off=2 * co, c_seg_stride=os_,
             ldc=r * co)
    torch.cuda.synchronize()
    assert torch.equal(out2.cpu()[:, 2:2 + T * r], got)
    with pytest.raises(hip.SoproHipError):
        hip.seanet_up128(xd, Wd, bd, out2, B=B, T=T, x_seg_stride=xs, out_seg_stride=os_, passes=2)


@pytest.mark.parametrize("N,K,glu", [(1536, 384, False), (384, 1536, False), (2049, 384, False), (768, 384, True)])
def test_skinny_packed_weights_are_the_same_function(N, K, glu):
    """sopro_pack_skinny_w only changes where the kernel finds a weight: bit-identical outputs (ragged last tile, K slices, GLU tail)."""
    B, D = 19, 384
    X, W, b = rnd(B, K, seed=801), rnd(N, K, seed=802, scale=K ** -0.5), rnd(N, seed=803)
    Wd, Wp = dev(W), hip.pack_skinny_w(dev(W), glu=glu)
    outs = []
    for Wx in (Wd, Wp):
        if glu:
            L, k, dil = 5, 3, 2
            ring = dev(rnd(L, B, D, seed=804))
            step = torch.tensor([3], dtype=torch.int32, device=DEV)
            Y = torch.full((B, D), float("nan"), device=DEV)
            hip.skinny(dev(X), Wx, Y, B=B, N=N, K=K, rms_norm=True, eps=1e-6, bias=dev(b), epilogue=hip.EPI_GLU_DW, ring=ring,
                       dw_w=dev(rnd(k, D, seed=805)), dw_b=dev(rnd(D, seed=806)), step=step, ring_len=L, dil=dil, ksize=k, ring_bcap=B)
            outs.append((Y.cpu(), ring.cpu()))
        elif K > 384:
            Y = torch.full((K // 384, B, N), float("nan"), device=DEV)
            R = dev(rnd(B, N, seed=807))
            hip.skinny(dev(X), Wx, Y, B=B, N=N, K=K, bias=dev(b), epilogue=hip.EPI_RES, R=R, ksplit=True, y_part_stride=B * N)
            outs.append((Y.cpu(),))
        else:
            Y = torch.full((B, N), float("nan"), device=DEV)
            hip.skinny(dev(X), Wx, Y, B=B, N=N, K=K, rms_norm=True, eps=1e-6, bias=dev(b), epilogue=hip.EPI_GELU)
            outs.append((Y.cpu(),))
    torch.cuda.synchronize()
    for a, c in zip(outs[0], outs[1]):
        assert bool(torch.isfinite(a).all()) and torch.equal(a, c)
    with pytest.raises(hip.SoproHipError):  # a packed weight knows its shape
        hip.skinny(dev(X), Wp, torch.empty(B, N, device=DEV), B=B, N=N + 16, K=K)


# ------------------------------------------------------------------------------------------ sampler
class _SamplerRig:
    def __init__(self, B, Tar=64, D=384, V=2048):
        z = lambda *s, dt=torch.float32: torch.zeros(*s, dtype=dt, device=DEV)  # noqa: E731
        self.B, self.Tar, self.D, self.V = B, Tar, D, V
        self.x, self.cond, self.emb = z(B, D), dev(rnd(B, Tar, D, seed=90)), dev(rnd(V * 2 + 1, D, seed=91))
        self.hist, self.ctr = z(B, Tar, dt=torch.int32), z(8, dt=torch.int32)
        self.first_eos, self.stop_t, self.params = z(B, dt=torch.int32), z(B, dt=torch.int32), z(8)
        self.recent = z(B, 64, dt=torch.int32)
        self.nonce = z(B, dt=torch.int32)
        self.row_step = z(B, dt=torch.int32)
        self.key = torch.tensor([7, 0], dtype=torch.int32, device=DEV)  # Philox key (seed) in device memory
        st = hip.ArState()
        st.nonce = self.nonce.data_ptr()
        st.row_step, st.key = self.row_step.data_ptr(), self.key.data_ptr()
        st.x_cur, st.cond, st.emb, st.hist = self.x.data_ptr(), self.cond.data_ptr(), self.emb.data_ptr(), self.hist.data_ptr()
        st.step, st.n_stopped = self.ctr.data_ptr(), self.ctr.data_ptr() + 8
        st.first_eos, st.stop_t, st.params = self.first_eos.data_ptr(), self.stop_t.data_ptr(), self.params.data_ptr()
        st.recent = self.recent.data_ptr()
        st.seed, st.B, st.D, st.Tar, st.max_steps, st.V, st.bos_row = 7, B, D, Tar, Tar, V, 2 * V
        self.st = st

    def set_params(self, top_p, temp, anti, min_gen=12, rec_p=0.85, rec_t=1.2, rep=1.1, top_k=50.0):
        self.params.copy_(torch.tensor([top_p, temp, 1.0 if anti else 0.0, rec_p, rec_t, rep, float(top_k), float(min_gen)]))


def test_sampler_greedy_penalty_temperature_and_bookkeeping():
    B, V1 = 6, 2049
    rig = _SamplerRig(B)
    rig.set_params(0.0, 0.8, False, min_gen=3)
    hip.ar_init(rig.st)
    torch.cuda.synchronize()
    close(rig.x, rig.cond[:, 0].cpu() + rig.emb[2 * 2048].cpu(), 0.0, "ar_init x")
    hist = [[] for _ in range(B)]
    for t in range(8):
        lg = rnd(B, V1, seed=200 + t, scale=3.0)
        if t == 1:
            lg[2, 2048] = 50.0  # EOS before min_gen: recorded as first_eos, row keeps going
        if t == 5:
            lg[4, 2048] = 50.0  # EOS after min_gen: row stops
        hip.ar_sample(rig.st, dev(lg), V1)
        torch.cuda.synchronize()
        got = rig.hist[:, t].cpu().tolist()
        for b in range(B):
            want = int(torch.argmax(O.penalised_logits(lg[b], hist[b], 0.8, 1.1)))
            assert got[b] == want, (t, b, got[b], want)
            hist[b].append(want)
        assert int(rig.ctr[0]) == t + 1
        ref_x = rig.cond[:, t + 1].cpu() + rig.emb.cpu()[torch.tensor(got)]
        close(rig.x, ref_x, 0.0, "next input")
    assert rig.first_eos.cpu().tolist() == [-1, -1, 1, -1, 5, -1]
    assert rig.stop_t.cpu().tolist() == [-1, -1, -1, -1, 5, -1]
    assert int(rig.ctr[2]) == 1


def test_sampler_topk_topp_support_and_frequencies():
    """Stochastic draws: every token must lie in the reference's kept set and the head token's frequency
    must match its probability (distributional parity; the CPU RNG stream cannot be matched)."""
    B, V1, steps = 32, 2049, 24
    rig = _SamplerRig(B, Tar=steps + 1)
    rig.set_params(0.9, 1.05, False)
    hip.ar_init(rig.st)
    base = rnd(V1, seed=300, scale=2.0)
    lg = base[None].repeat(B, 1)
    hits, expect = 0, 0.0
    hist = [[] for _ in range(B)]
    for t in range(steps):
        hip.ar_sample(rig.st, dev(lg), V1)
        torch.cuda.synchronize()
        got = rig.hist[:, t].cpu().tolist()
        for b in range(B):
            sp, si, forced = O.sampling_distribution(lg[b], hist[b], 0.9, 1.05)
            kept = set(si[sp > 0].tolist())
            assert got[b] in kept, (t, b, got[b])
            hits += int(got[b] == int(si[0]))
            expect += float(sp[0])
            hist[b].append(got[b])
    n = B * steps
    p = expect / n
    assert abs(hits / n - p) < 5 * math.sqrt(p * (1 - p) / n) + 0.02, (hits / n, p)


@pytest.mark.parametrize("top_p,temp,top_k,scale", [(0.9, 1.05, 50, 2.0), (0.6, 0.9, 50, 4.0), (1.0, 1.0, 8, 1.0), (0.97, 1.3, 64, 0.3),
                                                     (0.9, 1.0, 50, -1.0)])
def test_sampler_whole_distribution_matches_reference(top_p, temp, top_k, scale):
    """Empirical distribution of ~6000 draws against sample_token's distribution (src/sopro/sampling.py:52-80), token by
    token (5 sigma + 0.004), nothing outside the kept set; flat-ish logits put >64 candidates through the general ranking
    path, top_k = 8 / 64 exercise the bound at both ends.  No repetition penalty here, so the distribution is fixed."""
    B, V1, steps = 64, 2049, 96
    rig = _SamplerRig(B, Tar=steps + 1)
    rig.set_params(top_p, temp, False, rep=1.0, top_k=top_k)
    rig.nonce.copy_(torch.arange(B, dtype=torch.int32) * 7919 + 13)
    hip.ar_init(rig.st)
    base = rnd(V1, seed=310, scale=abs(scale))
    if scale < 0:  # a crowded head: 200 near-equal logits above the rest -> more than 128 candidates, the general ranking path
        base[:200] = 10.0 - 1e-3 * torch.arange(200)
    else:
        base[7] = base[11] = float(base.max()) + 0.5  # an exact tie at the head: lower index first
    lg = dev(base[None].repeat(B, 1))
    for _ in range(steps):
        hip.ar_sample(rig.st, lg, V1)
    torch.cuda.synchronize()
    got = rig.hist[:, :steps].cpu().reshape(-1)
    sp, si, forced = O.sampling_distribution(base, [], top_p, temp, top_k=top_k, repetition_penalty=1.0)
    assert forced is None
    p = torch.zeros(V1, dtype=torch.float64)
    p[si] = sp.double()
    n = got.numel()
    freq = torch.bincount(got.long(), minlength=V1).double() / n
    assert float(freq[p == 0].sum()) == 0.0, "a token outside the reference's kept set was drawn"
    tol = 5.0 * torch.sqrt(p * (1 - p) / n) + 0.004
    worst = ((freq - p).abs() - tol).max()
    assert float(worst) <= 0.0, (float(worst), int(((freq - p).abs() - tol).argmax()))
    assert int((p > 0).sum()) >= 2


def test_sampler_matches_reference_draws_with_history():
    """ar_sample_kernel against 20 000 seeded draws of THE REFERENCE's sample_token per case (tests/golden/sampler.npz, made by
    tests/golden/make_golden_sampler.py from /root/reference/src/sopro/sampling.py:24-93): every case has a 60-token history
    (repetition penalty over the set of the last 50), so the whole temperature -> penalty -> softmax -> top-k -> top-p order
    is compared.  Two-sample test per token (5 sigma of the difference of two empirical frequencies + 0.003), nothing outside
    the reference's drawn set beyond what 20 000 draws can miss."""
    g = golden("sampler")
    n_ref = int(g["n_draws"])
    B, V1, steps = 256, 2049, 48
    for ci, (top_p, temp, top_k, _scale) in enumerate(g["cases"].tolist()):
        logits, hist = _t(g[f"logits{ci}"]).float(), g[f"hist{ci}"].tolist()
        ref_freq = _t(g[f"counts{ci}"]).double() / n_ref
        rig = _SamplerRig(B, Tar=steps + 1)
        rig.set_params(top_p, temp, False, rep=1.1, top_k=int(top_k))
        rig.nonce.copy_(torch.arange(B, dtype=torch.int32) * 104729 + 17 + ci)
        hip.ar_init(rig.st)
        window = torch.full((64,), -1, dtype=torch.int32)
        for j in range(min(64, len(hist))):
            window[j] = hist[-1 - j]  # slot j = the token sampled j + 1 frames ago
        rec0 = dev(window[None].repeat(B, 1))
        lg = dev(logits[None].repeat(B, 1))
        for _ in range(steps):
            rig.recent.copy_(rec0)  # the same history for every draw (the kernel appends its token)
            hip.ar_sample(rig.st, lg, V1)
        torch.cuda.synchronize()
        got = rig.hist[:, :steps].cpu().reshape(-1).long()
        n = got.numel()
        freq = torch.bincount(got, minlength=V1).double() / n
        sp, si, forced = O.sampling_distribution(logits, hist, top_p, temp, top_k=int(top_k), repetition_penalty=1.1)
        p = torch.zeros(V1, dtype=torch.float64)
        p[si] = sp.double()
        assert float(freq[p == 0].sum()) == 0.0, (ci, "a token outside the kept set was drawn")
        pool = (freq * n + ref_freq * n_ref) / (n + n_ref)
        tol = 5.0 * torch.sqrt(pool * (1 - pool) * (1.0 / n + 1.0 / n_ref)) + 0.003
        worst = ((freq - ref_freq).abs() - tol).max()
        assert float(worst) <= 0.0, (ci, float(worst), int(((freq - ref_freq).abs() - tol).argmax()))


def test_sampler_nonce_changes_the_take_and_pins_it():
    """Same (seed, nonce) -> the same draws; another nonce -> another take (every call used to replay one sequence)."""
    B, V1, steps = 8, 2049, 16
    base = rnd(V1, seed=320, scale=1.5)
    lg = dev(base[None].repeat(B, 1))
    runs = []
    for nonce in (5, 5, 6):
        rig = _SamplerRig(B, Tar=steps + 1)
        rig.set_params(0.95, 1.0, False, rep=1.0)
        rig.nonce.fill_(nonce)
        hip.ar_init(rig.st)
        for _ in range(steps):
            hip.ar_sample(rig.st, lg, V1)
        torch.cuda.synchronize()
        runs.append(rig.hist[:, :steps].cpu())
    assert torch.equal(runs[0], runs[1])
    assert not torch.equal(runs[0], runs[2])
    assert len({tuple(r.tolist()) for r in runs[0]}) > 1  # rows differ too (the row index is part of the counter)


def test_sampler_key_in_device_memory_and_run_to_run_determinism():
    """(1) the Philox key is read from device memory, so a recorded frame graph follows a new seed; (2) candidate
    positions no longer depend on which wave arrives first: the same (key, nonce) gives the same tokens run after run, also on
    the many-candidates path and with logits that put the top-p cut / the draw on close calls."""
    B, V1, steps = 32, 2049, 24
    for scale in (1.5, 0.02, -1.0):
        base = rnd(V1, seed=330, scale=abs(scale))
        if scale < 0:
            base[:200] = 10.0 - 1e-3 * torch.arange(200)  # > 128 candidates: the general ranking path
        lg = dev(base[None].repeat(B, 1))
        runs = []
        rig = _SamplerRig(B, Tar=steps + 1)
        rig.set_params(0.9, 1.05, False)
        s = torch.cuda.Stream()
        with torch.cuda.stream(s):
            hip.capture_begin()
            hip.ar_sample(rig.st, lg, V1)
            g = hip.capture_end()  # ONE recorded launch serves every key
            for key in (7, 7, 7, 8):
                rig.key.copy_(torch.tensor([key, 0], dtype=torch.int32), non_blocking=False)
                hip.ar_init(rig.st)
                for _ in range(steps):
                    g.launch()
                s.synchronize()
                runs.append(rig.hist[:, :steps].cpu())
        assert torch.equal(runs[0], runs[1]) and torch.equal(runs[0], runs[2]), scale
        assert not torch.equal(runs[0], runs[3]), scale


def test_sampler_anti_loop_detection():
    """repeated_tail / streak >= 8 switch that frame to the recovery parameters (model.py:274-279).
    Recovery top_p is set to 0 here so that a detected loop shows up as an exact arg-max."""
    B, V1 = 4, 2049
    rig = _SamplerRig(B, Tar=40)
    rig.set_params(1.0, 1.0, True, rec_p=0.0, rec_t=1.0)
    hip.ar_init(rig.st)
    h = torch.zeros(B, 40, dtype=torch.int32)
    h[0, :6] = torch.tensor([5, 6, 7, 5, 6, 7])       # repeated tail n=3
    h[1, :6] = torch.tensor([5, 6, 7, 5, 6, 8])       # no loop
    h[2, :10] = torch.tensor([1] + [9] * 9)           # 9 equal tokens: streak 8
    h[3, :10] = torch.tensor([1, 2] + [9] * 8)        # only 8 equal: streak 7
    flat = 0.01 * rnd(V1, seed=400)
    flags = []
    for row, L in ((0, 6), (1, 6), (2, 10), (3, 10)):
        rig.hist.copy_(h)
        rec = torch.full((B, 64), -1, dtype=torch.int32)
        for r_, L_ in ((0, 6), (1, 6), (2, 10), (3, 10)):
            rec[r_, :L_] = h[r_, :L_].flip(0)  # slot j = token sampled j+1 frames ago
        rig.recent.copy_(rec)
        rig.ctr.zero_()
        rig.ctr[0] = L
        rig.row_step.fill_(L)  # the sampler's own (per-row) frame counter
        lg = flat[None].repeat(B, 1)
        hip.ar_sample(rig.st, dev(lg), V1)
        torch.cuda.synchronize()
        want = int(torch.argmax(O.penalised_logits(flat, h[row, :L].tolist(), 1.0, 1.1)))
        flags.append(int(rig.hist[row, L]) == want)
        assert rig.recent[row, : L + 1].cpu().tolist() == [int(rig.hist[row, L])] + h[row, :L].flip(0).tolist()
    assert flags[0] and flags[2], flags
    assert not (flags[1] and flags[3]), flags  # near-uniform logits over the top-50: a chance arg-max on both rows is ~4e-4 (and fixed by the Philox seed)


def test_graph_capture_and_replay():
    x = dev(rnd(64, seed=500))
    out = torch.zeros(64, device=DEV)
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        hip.capture_begin()
        hip.tanh_affine(x, out, 1.0, 2.0, 64)
        hip.tanh_affine(out, out, 0.0, 1.0, 64)
        g = hip.capture_end()
        assert float(out.abs().max()) == 0.0  # capture records, it does not run
        g.launch()
        g.launch()
    s.synchronize()
    close(out, torch.tanh(1.0 + 2.0 * torch.tanh(x.cpu())), 1e-6, "graph")


def _to_split_form(x):
    """fp32 rows [..., C] (C % 32 == 0) -> the split form a producer's c_mode 1 / 2 writes (without the ELU): every 32 channels =
    [32 hi bf16 | 32 lo bf16], returned as the fp32-typed tensor of the same shape whose BYTES those are."""
    sh = x.shape
    g = x.reshape(-1, sh[-1] // 32, 32).float()
    hi = g.to(torch.bfloat16)
    lo = (g - hi.float()).to(torch.bfloat16)
    return torch.cat([hi, lo], dim=-1).contiguous().view(torch.float32).reshape(sh)


@pytest.mark.parametrize("K", [1024, 2048 + 32])
def test_gemm_long_k_form_is_the_same_function(K):
    """Round 6 (csrc/gemm_8p.hip): 256 x 256 tiles, both operands split in memory and staged by LDS-DMA, against the 128 x 128 tile
    kernel on the same split-form A: the same products in the same order per output element -> bit-identical for every output mode
    (fp32 rows; ELU + split form; raw + ELU split; raw + ELU fp32; the tile kernel unsplit in K), with a ragged last row tile, several tiles per CU, an odd K-tile
    count, and segments of overlapping rows (lda < K: a transposed convolution's row windows).  Both within the three-pass error class
    of the float64 product."""
    lib = hip.load()
    M, N = 3073, 512  # (>= 96 tiles of 128 x 128: the tile kernel runs unsplit in K, as it does on the decoder's shapes)
    A, W, b = rnd(M, K, seed=191), rnd(N, K, seed=192, scale=K ** -0.5), rnd(N, seed=193)
    As = dev(_to_split_form(A))
    Wp = hip.pack_w_bf16x3(dev(W), rows=True)
    ref = A.double() @ W.double().t() + b.double()
    mag = A.double().abs() @ W.double().abs().t() + b.double().abs()
    g = hip.GemmArgs()
    x = hip.SplitExt()
    g.M, g.N, g.K, g.prologue, g.epilogue = 25600, N, K, 0, 0
    x.a_format = 1
    assert lib.sopro_gemm_8p_takes(C.byref(g), C.byref(x)) == 1   # the decoder's shapes go there by themselves ...
    g.M = M
    assert lib.sopro_gemm_8p_takes(C.byref(g), C.byref(x)) == 0   # ... a handful of tiles does not (here it is forced: long_k=True)

    def run(long_k, c_mode):
        Cm = torch.full((M, N), float("nan"), device=DEV)
        C2 = torch.full((M, N), float("nan"), device=DEV) if c_mode in (2, 4) else None
        hip.gemm(As, Wp, Cm, M=M, N=N, K=K, bias=dev(b), a_split=True, c_mode=c_mode, C2=C2, long_k=long_k)
        torch.cuda.synchronize()
        return Cm.cpu(), (C2.cpu() if C2 is not None else None)

    for c_mode in (0, 1, 2, 4):
        (c_t, c2_t), (c_l, c2_l) = run(False, c_mode), run(True, c_mode)
        assert torch.equal(c_t.view(torch.int32), c_l.view(torch.int32)), c_mode
        if c2_t is not None:
            assert torch.equal(c2_t.view(torch.int32), c2_l.view(torch.int32)), c_mode
        if c_mode in (0, 2, 4):
            assert bool(torch.isfinite(c_l).all()) and float(((c_l.double() - ref).abs() / mag).max()) < 2e-5
        if c_mode == 4:
            close(c2_l, F.elu(c_l), 1e-6, "ELU copy")
    # row windows: T rows of ci channels per utterance, one zero row in front, window = 2 rows (K = 2 ci), two utterances
    ci, T, Bn = K // 2, 1600, 2
    if ci % 32 == 0:
        xw = torch.zeros(Bn, 1 + T, ci)
        xw[:, 1:] = rnd(Bn, T, ci, seed=194)
        xs = dev(_to_split_form(xw))
        outs = []
        for long_k in (False, True):
            Cm = torch.full((Bn, 2 + T * 2, N // 2), float("nan"), device=DEV)
            Cm[:, :2] = 0.0
            hip.gemm(xs, Wp, Cm, M=Bn * T, N=N, K=K, lda=ci, bias=dev(b), rows_per_seg=T, a_seg_stride=(1 + T) * ci, a_split=True,
                     c_off=2 * (N // 2), c_seg_stride=(2 + 2 * T) * (N // 2), ldc=N, long_k=long_k)
            torch.cuda.synchronize()
            outs.append(Cm.cpu())
        assert bool(torch.isfinite(outs[1]).all()) and torch.equal(outs[0], outs[1])
        win = torch.cat([xw[:, :-1], xw[:, 1:]], dim=-1).reshape(Bn * T, K)
        wref = (win.double() @ W.double().t() + b.double()).reshape(Bn, T * 2, N // 2)
        assert float((outs[1][:, 2:].double() - wref).abs().max()) < 2e-4
    with pytest.raises(hip.SoproHipError):  # fp32 rows are not what the DMA form stages
        hip.gemm(dev(A), Wp, torch.empty(M, N, device=DEV), M=M, N=N, K=K, long_k=True)
    with pytest.raises(hip.SoproHipError):  # no rows operand
        hip.gemm(As, hip.pack_w_bf16x3(dev(W)), torch.empty(M, N, device=DEV), M=M, N=N, K=K, a_split=True, long_k=True)
