"""CPU restatement of what attn_mfma_split_kernel computes (sopro_amd/csrc/attention_mfma.hip), to pin its error budget and the
lazy softmax reference without a GPU: operands as two bf16 pieces (x = hi + lo, both round-to-nearest), products without the
lo*lo term, scores in the exp2 domain (Q carries scale * log2 e), 32-key tiles, a running reference that only moves when some
query outgrows it by 2^8.  The GPU test of the kernel itself is tests/test_gpu_ops.py::test_attention_window_split_bf16_form."""
import math

import torch


def _split(x):
    hi = x.to(torch.bfloat16).to(torch.float32)
    lo = (x - hi).to(torch.bfloat16).to(torch.float32)
    return hi, lo


def _mm3(a, b):
    """a @ b with both operands as two bf16 pieces, three of the four piece products, fp32 accumulation (emulated in fp64 sums
    of exactly representable products: the accumulation order of the MFMAs is not what this test is about)."""
    ah, al = _split(a)
    bh, bl = _split(b)
    d = torch.float64
    return (al.to(d) @ bh.to(d) + ah.to(d) @ bl.to(d) + ah.to(d) @ bh.to(d)).to(torch.float32)


def split_window_attention(q, k, v, window, past, lazy=True, thresh=8.0):
    """q [N, dh], k / v [Tk, dh] of one head; queries at positions past .. past+N-1, keys at past+N-Tk .. past+N-1."""
    N, dh = q.shape
    Tk = k.shape[0]
    kpos0 = past + N - Tk
    qs = q * (dh ** -0.5 * 1.4426950408889634)
    out = torch.zeros(N, dh)
    stats = {"max_p": 0.0, "rescales": 0, "tiles": 0}
    for q0 in range(0, N, 32):
        qt = qs[q0:q0 + 32]
        nq = qt.shape[0]
        qabs = past + q0 + torch.arange(nq)
        m = torch.full((nq,), -math.inf)
        l = torch.zeros(nq)
        o = torch.zeros(nq, dh)
        k_first = max(int(qabs[0]) - window + 1 - kpos0, 0)
        k_last = min(int(qabs[-1]) - kpos0, Tk - 1)
        for k0 in range((k_first // 32) * 32, k_last + 1, 32):
            kt, vt = k[k0:k0 + 32], v[k0:k0 + 32]
            kabs = kpos0 + k0 + torch.arange(kt.shape[0])
            s = _mm3(qt, kt.t())  # [nq, nk], exp2 domain
            ok = (kabs[None, :] <= qabs[:, None]) & (kabs[None, :] > qabs[:, None] - window)
            s = torch.where(ok, s, torch.tensor(-math.inf))
            mx = s.max(dim=1).values
            grow = (mx > m + thresh) | torch.isinf(m)
            stats["tiles"] += 1
            if (not lazy) or bool(grow.any()):  # the kernel's wave-uniform branch (one wave = these 32 queries)
                m_new = torch.maximum(m, mx)
                alpha = torch.where(torch.isinf(m_new) | (m_new == m), torch.ones(nq), torch.exp2(m - m_new))
                l, o, m = l * alpha, o * alpha[:, None], m_new
                stats["rescales"] += 1
            p = torch.where(torch.isinf(s), torch.zeros(()), torch.exp2(s - m[:, None]))
            p = torch.nan_to_num(p, nan=0.0)  # rows that have seen nothing yet (m = -inf, s = -inf)
            stats["max_p"] = max(stats["max_p"], float(p.max()))
            l = l + p.sum(dim=1)
            o = o + _mm3(p, vt)
        out[q0:q0 + nq] = o / l[:, None]
    return out, stats


def reference(q, k, v, window, past):
    N, dh = q.shape
    Tk = k.shape[0]
    pos = past + torch.arange(N)
    kpos = past + N - Tk + torch.arange(Tk)
    vis = (kpos[None, :] <= pos[:, None]) & (kpos[None, :] > pos[:, None] - window)
    s = (q.double() @ k.double().t()) / math.sqrt(dh)
    s = s.masked_fill(~vis, -math.inf)
    return (torch.softmax(s, -1) @ v.double()).float()


def test_split_window_attention_error_budget_and_lazy_reference():
    g = torch.Generator().manual_seed(5)
    for N, win, past, scale in [(400, 250, 0, 1.0), (96, 40, 1000, 1.0), (130, 250, 300, 4.0), (33, 17, 5, 1.0)]:
        Tk = N + min(past, win - 1)
        q, k, v = (torch.randn(n, 64, generator=g) * sc for n, sc in ((N, scale), (Tk, scale), (Tk, 1.0)))
        ref = reference(q, k, v, win, past)
        lazy, st = split_window_attention(q, k, v, win, past, lazy=True)
        eager, _ = split_window_attention(q, k, v, win, past, lazy=False)
        top = float(ref.abs().max())
        # 16-bit operands: a score is off by ~2^-17 |q||k| scale (random signs), a weight by that much relatively - the budget
        # grows with the operands' energy (scale^2 here); unit-variance rows are the GPU test's case and its 3e-5 bound
        tol = 3e-5 * scale * scale * top
        assert float((lazy - ref).abs().max()) < tol, (N, win, past)
        assert float((eager - ref).abs().max()) < tol
        # the lazy reference changes rounding only: weights stay below 2^8 (exact in fp32, well inside bf16's range) ...
        assert st["max_p"] <= 2.0 ** 8 * (1 + 1e-6)
        assert float((lazy - eager).abs().max()) < 1e-5 * scale * scale * top
        # ... and it does skip most rescales once the running maxima have settled (what it is for)
        if N >= 96 and scale == 1.0:
            assert st["rescales"] < 0.8 * st["tiles"], st


def test_two_piece_products_carry_sixteen_bits():
    g = torch.Generator().manual_seed(6)
    a, b = torch.randn(64, 256, generator=g), torch.randn(256, 48, generator=g)
    exact = a.double() @ b.double()
    bound = (a.abs().double() @ b.abs().double())
    assert float(((_mm3(a, b).double() - exact).abs() / bound).max()) < 2.0 ** -15
    one = (a.to(torch.bfloat16).double() @ b.to(torch.bfloat16).double())
    assert float(((one - exact).abs() / bound).max()) > 2.0 ** -12  # one piece is what bf16 mode accepts, not the waveform path
