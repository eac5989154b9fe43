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


@pytest.fixture(scope="module")
def tts_bf16(cfg, sopro_np_noeos, mimi_np):
    from sopro_amd import SoproTTS
    return SoproTTS.from_weights(cfg, sopro_np_noeos, mimi_np, FakeTok(), device="cuda:0", precision="bf16")


def test_bf16_mode_quality_against_the_fp32_reference(tts_bf16, cfg, mc, w_noeos):
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
    # codebook 0 (the AR loop) is fp32 in this mode: generation from text must reproduce the reference's codebook-0 tokens
    toks = tts.model.generate_tokens(torch.from_numpy(g["ids"]), ref, max_frames=int(g["max_frames"]), style_strength=1.0,
                                     top_p=0.0, temperature=1.0, anti_loop=False)
    cb0_equal = bool(torch.equal(toks[:, 0].cpu(), want[:, 0]))
    print(f"\\nbf16 mode vs fp32 reference (full200): cond_ar max err {cond_err:.2e}; refined-token agreement {agree:.4f} "
          f"({n_off} of {T * 31} off the fp32 arg-max, worst fp32 logit gap {gap:.3f}); waveform max err {werr:.3e} of peak, SNR {snr:.1f} dB; "
          f"codebook 0 equal: {cb0_equal}")
    assert agree > 0.80 and werr < 0.05 and snr > 30.0
