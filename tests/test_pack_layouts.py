"""Host logic: the weight repacking of sopro_amd/pack.py turns Conv1d / ConvTranspose1d / GLU into the
row-window contractions the HIP GEMM computes.  Checked on the CPU by emulating the kernel's addressing
(``as_strided`` windows over zero-padded channels-last buffers) against torch's own conv ops."""
import torch
import torch.nn.functional as F

from sopro_amd import pack


def _windows(buf_rows, n_rows, K, lda, off):
    """Row m = K consecutive floats starting at off + m*lda of the flattened buffer."""
    flat = buf_rows.reshape(-1)
    return torch.as_strided(flat, (n_rows, K), (lda, 1), off)


def test_conv1d_as_row_window_contraction():
    torch.manual_seed(0)
    T, ci, co, k = 11, 8, 12, 7
    x = torch.randn(T, ci)
    w = torch.randn(co, ci, k)
    b = torch.randn(co)
    ref = F.conv1d(F.pad(x.t().unsqueeze(0), (k - 1, 0)), w, b)[0].t()  # causal, HF:modeling_mimi.py:327-347
    buf = torch.cat([torch.zeros(k - 1, ci), x], dim=0)
    A = _windows(buf, T, k * ci, ci, 0)
    out = A @ pack.pack_conv1d(w).t() + b
    assert torch.allclose(out, ref, atol=1e-5)


def test_convtranspose1d_as_row_window_contraction():
    torch.manual_seed(1)
    for s in (2, 4, 5, 6, 8):
        T, ci, co = 9, 8, 4
        x = torch.randn(T, ci)
        w = torch.randn(ci, co, 2 * s)
        b = torch.randn(co)
        y = F.conv_transpose1d(x.t().unsqueeze(0), w, b, stride=s)[0]
        ref = y[:, : y.shape[1] - s].t()  # trim k - s on the right (HF:modeling_mimi.py:399-405) -> [T*s, co]
        wp, bp = pack.pack_convtr1d(w, b, s)
        buf = torch.cat([torch.zeros(1, ci), x], dim=0)
        A = _windows(buf, T, 2 * ci, ci, 0)
        out = (A @ wp.t() + bp).reshape(T * s, co)  # row t holds samples t*s .. t*s+s-1, channels-last
        assert torch.allclose(out, ref, atol=1e-5), s


def test_glu_packing_pairs_value_and_gate_rows():
    torch.manual_seed(2)
    d, k = 64, 16
    w = torch.randn(2 * d, k)
    b = torch.randn(2 * d)
    x = torch.randn(5, k)
    y = x @ w.t() + b
    ref = y[:, :d] * torch.sigmoid(y[:, d:])
    wp, bp = pack.pack_glu(w, b)
    yp = x @ wp.t() + bp
    out = torch.empty(5, d)
    for g in range(d // 32):  # the GEMM epilogue: columns [64g, 64g+32) gate-multiplied by [64g+32, 64g+64)
        out[:, 32 * g: 32 * g + 32] = yp[:, 64 * g: 64 * g + 32] * torch.sigmoid(yp[:, 64 * g + 32: 64 * g + 64])
    assert torch.allclose(out, ref, atol=1e-6)


def test_dw_taps_and_tables():
    w = torch.arange(24.0).reshape(4, 1, 6)
    p = pack.pack_dw(w)
    assert p.shape == (6, 4) and p[2, 3] == w[3, 0, 2]
    pe = pack.sinusoid_table(16, 8)
    from oracle import sopro_oracle as O
    assert torch.equal(pe, O.sinusoid(torch.arange(16), 8))
    c, s = pack.rope_tables(10, 64, 10000.0)
    oc, os_ = O.rope_cos_sin(torch.arange(10), 64, 10000.0)
    assert torch.equal(c, oc[:, :32]) and torch.equal(s, os_[:, :32])


def test_pack_covers_reference_checkpoint_names(cfg, mc, sopro_np, mimi_np):
    ps = pack.pack_sopro(sopro_np, cfg)
    pm = pack.pack_mimi(mimi_np, mc)
    assert ps["ar.blocks.0.glu.w"].shape == (768, 384) and ps["nar.blocks.5.dw.w"].shape == (11, 384)
    assert ps["ar.x_attns.1.kv.w"].shape == (768, 384) and ps["ar.x_attns.5.gate_scale"].shape == (384,)
    assert pm["codebooks"].shape == (32 * 2048, 256) and pm["rvq_proj.w"].shape == (512, 512)
    assert pm["sea.conv0.w"].shape == (1024, 7 * 512)
    assert pm["sea.up0.w"].shape == (8 * 512, 2 * 1024) and pm["sea.up3.w"].shape == (4 * 64, 2 * 128)
    assert pm["sea.res3.c1.w"].shape == (32, 3 * 64) and pm["sea.final.w"].shape == (3, 64)


def test_unfolded_key_query_operands_reproduce_the_folded_scores():
    """Round 4 (sopro_ar_frame.k_unfold): the query of a text cross-attention block emitted by the feed-forward launches in front of
    it - q_raw = qa.w @ out + q.b + qu.w @ u with the operands pack_sopro folds in float64 - must be Wq' x for x = out + b2 + W2 u,
    and <q_raw_h, K_h[k]> must be the folded score <x * w_nq, K'_h[k]> (reference: src/sopro/nn/text.py:85-132, blocks.py:158-162)."""
    import numpy as np
    from sopro_amd.config import SoproTTSConfig
    from sopro_amd.pack import pack_sopro
    from sopro_amd.weights import synth_sopro_weights

    cfg = SoproTTSConfig()
    wn = synth_sopro_weights(cfg, 512, 3)
    p = pack_sopro(wn, cfg)
    g = torch.Generator().manual_seed(5)
    D, H = 384, 4
    dh = D // H
    for i in cfg.ar_xattn_layers:
        pa = f"ar.x_attns.{i}"
        out, u = torch.randn(7, D, generator=g).double(), torch.randn(7, 4 * D, generator=g).double()
        W2, b2 = torch.from_numpy(wn[f"ar.blocks.{i}.ff.3.weight"]).double(), torch.from_numpy(wn[f"ar.blocks.{i}.ff.3.bias"]).double()
        x = out + b2 + u @ W2.t()
        wq = torch.from_numpy(wn[pa + ".q_proj.weight"]).double() * torch.from_numpy(wn[pa + ".nq.weight"]).double()[None, :]
        q_ref = x @ wq.t()
        q_new = out @ p[pa + ".qa.w"].double().t() + p[pa + ".q.b"].double() + u @ p[pa + ".qu.w"].double().t()
        assert float((q_new - q_ref).abs().max()) < 1e-5 * float(q_ref.abs().max())
        K = torch.randn(7, 11, D, generator=g).double()  # unfolded keys, head h in columns 96 h ..
        for h in range(H):
            Kp = K[:, :, h * dh:(h + 1) * dh] @ p[pa + ".q.wT"][h].double().t()  # folded K'_h = K_h Wq_h (norm weight folded): [7, 11, D]
            s_folded = torch.einsum("bd,bkd->bk", x, Kp)
            s_unfolded = torch.einsum("bd,bkd->bk", q_ref[:, h * dh:(h + 1) * dh], K[:, :, h * dh:(h + 1) * dh])
            assert float((s_folded - s_unfolded).abs().max()) < 1e-5 * float(s_folded.abs().max())
