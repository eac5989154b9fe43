"""The last SEANet level as ONE kernel (csrc/seanet_uptail.hip; HF:modeling_mimi.py:931-961, 408-447): the last transposed
convolution, the last residual block and the last layer without the 64-channel activation's round trip through memory.
Checked against torch's layers, against the two kernels it replaces (same operand rounding, another summation order in the
residual block), across tile partitions (bit-identical: the carried rows / warm-up tile are exact) and on bf16 rows."""
import pytest
import torch
import torch.nn.functional as F

from oracle import sopro_oracle as O
from sopro_amd import hip, pack

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
CI, CO, R = 128, 64, 4


def rnd(*shape, seed=0, scale=1.0):
    g = torch.Generator().manual_seed(seed)
    return torch.randn(*shape, generator=g) * scale


def dev(t):
    return t.to(DEV).contiguous()


def bf(t):
    return t.to(torch.bfloat16).to(torch.float32)


def _weights():
    wt, bt = rnd(CI, CO, 2 * R, seed=761, scale=0.06), rnd(CO, seed=762, scale=0.1)
    w1, b1 = rnd(32, 64, 3, seed=731, scale=0.07), rnd(32, seed=732, scale=0.1)
    w2, b2 = rnd(64, 32, 1, seed=733, scale=0.17), rnd(64, seed=734, scale=0.1)
    wf, bf_ = rnd(1, 64, 3, seed=735, scale=0.07), 0.03
    return wt, bt, w1, b1, w2, b2, wf, bf_


def _layers(x, wt, bt, w1, b1, w2, b2, wf, bf_, rd=lambda t: t):
    """torch's layers on the activated input x [B, T, 128] -> wav [B, 4 T]; ``rd`` rounds the matrix-core operands (bf16 mode)."""
    T = x.shape[1]
    h = F.conv_transpose1d(rd(x).transpose(1, 2), rd(wt), bt, stride=R)[..., :T * R]  # causal: trim the right tail
    y = O.causal_conv1d(rd(F.elu(h)), rd(w1), b1)
    y = O.causal_conv1d(rd(F.elu(y)), rd(w2), b2)
    return O.causal_conv1d(F.elu(h + y), wf, torch.tensor([bf_]))[:, 0]


def _device_args(wt, bt, w1, b1, w2, b2, wf):
    W, bias = pack.pack_convtr1d(wt, bt, R)  # [r*co, 2*ci], [r*co]
    return [dev(t) for t in (W, bias, pack.pack_conv1d(w1), b1, pack.pack_conv1d(w2), b2, wf[0].t())]


def _report(got, ref, what):
    """Where a mismatch sits: by utterance, by position within the 128-sample tile, by sample phase."""
    err = (got - ref).abs()
    if not torch.isfinite(err).all():
        bad = (~torch.isfinite(got)).nonzero()
        return f"{what}: {bad.shape[0]} non-finite samples, first at {bad[:4].tolist()}"
    peak = float(ref.abs().max())
    pos = err.argmax()
    b, s = int(pos // err.shape[1]), int(pos % err.shape[1])
    by_tile = torch.zeros(128)
    n = err.shape[1] // 128 * 128
    if n:
        by_tile = err[:, :n].reshape(err.shape[0], -1, 128).amax(dim=(0, 1))
    worst = by_tile.topk(min(6, 128))
    return (f"{what}: max err {float(err.max()):.3e} of peak {peak:.3e} at utterance {b} sample {s} (tile {s // 128}, in-tile {s % 128}); "
            f"worst in-tile positions {worst.indices.tolist()} {[f'{v:.1e}' for v in worst.values.tolist()]}; "
            f"first bad sample {int((err[b] > 1e-3 * peak).nonzero()[0]) if bool((err[b] > 1e-3 * peak).any()) else None}")


@pytest.mark.parametrize("B,T", [(1, 1), (2, 31), (2, 33), (3, 100), (2, 1000), (2, 4100)])
def test_uptail_three_pass_matches_the_layers_and_the_two_kernels(B, T):
    wt, bt, w1, b1, w2, b2, wf, bf_ = _weights()
    x = F.elu(rnd(B, T, CI, seed=760 + T))
    ref = _layers(x, wt, bt, w1, b1, w2, b2, wf, bf_)
    xs = (1 + T + 3) * CI
    xb = torch.zeros(B, 1 + T + 3, CI)
    xb[:, 1:1 + T] = x
    xd = dev(xb)
    Wd, bd, w1d, b1d, w2d, b2d, wfd = _device_args(wt, bt, w1, b1, w2, b2, wf)
    S = 4 * T
    lib, outs = hip.load(), []
    try:
        for tiles in (0, 1, 2, 5):
            lib.sopro_seanet_uptail_set_tiles(tiles)
            wav = torch.full((B, S + 5), float("nan"), device=DEV)
            hip.seanet_uptail(xd, Wd, bd, w1d, b1d, w2d, b2d, wfd, bf_, wav, B=B, T=T, x_seg_stride=xs, wav_seg_stride=S + 5)
            torch.cuda.synchronize()
            outs.append(wav.cpu())
    finally:
        lib.sopro_seanet_uptail_set_tiles(0)
    got = outs[0][:, :S]
    peak = float(ref.abs().max())
    assert bool(torch.isfinite(got).all()) and float((got - ref).abs().max()) <= 1e-4 * (peak + 1.0), _report(got, ref, "fused level vs torch")
    assert bool(torch.isnan(outs[0][:, S:]).all())  # nothing outside its samples
    for i, o in enumerate(outs[1:]):  # any partition of the tiles over workgroups: the same bits (carried rows, warm-up tile)
        assert torch.equal(o[:, :S], got), _report(o[:, :S], got, f"tiles setting #{i + 1} vs by-size")
    # the two kernels it replaces
    h = torch.zeros(B, 2 + S, CO, device=DEV)
    hip.seanet_up128(xd, Wd, bd, h, B=B, T=T, x_seg_stride=xs, out_seg_stride=(2 + S) * CO, out_off=2 * CO, passes=3)
    two = torch.full((B, S), float("nan"), device=DEV)
    hip.seanet_tail(h, w1d, b1d, w2d, b2d, wfd, bf_, two, B=B, T=S, h_seg_stride=(2 + S) * CO, wav_seg_stride=S)
    torch.cuda.synchronize()
    d = float((got - two.cpu()).abs().max())
    assert d <= 5e-6 * (peak + 1.0), _report(got, two.cpu(), "fused level vs the two kernels")


def test_uptail_one_pass_and_bf16_rows():
    B, T = 2, 700
    wt, bt, w1, b1, w2, b2, wf, bf_ = _weights()
    x = F.elu(rnd(B, T, CI, seed=790))
    Wd, bd, w1d, b1d, w2d, b2d, wfd = _device_args(wt, bt, w1, b1, w2, b2, wf)
    xs, S = (1 + T) * CI, 4 * T
    full = _layers(x, wt, bt, w1, b1, w2, b2, wf, bf_)
    peak = float(full.abs().max()) + 1.0
    # fp32 rows, operands rounded in flight
    xb = torch.zeros(B, 1 + T, CI)
    xb[:, 1:] = x
    wav = torch.full((B, S), float("nan"), device=DEV)
    hip.seanet_uptail(dev(xb), Wd, bd, w1d, b1d, w2d, b2d, wfd, bf_, wav, B=B, T=T, x_seg_stride=xs, wav_seg_stride=S, passes=1)
    ref1 = _layers(x, wt, bt, w1, b1, w2, b2, wf, bf_, rd=bf)
    assert float((wav.cpu() - ref1).abs().max()) <= 3e-3 * peak, _report(wav.cpu(), ref1, "one pass, fp32 rows")
    # bf16 rows
    x16 = torch.zeros(B, 1 + T, CI, dtype=torch.bfloat16)
    x16[:, 1:] = x.to(torch.bfloat16)
    outs = []
    try:
        for tiles in (0, 3):
            hip.load().sopro_seanet_uptail_set_tiles(tiles)
            wav16 = torch.full((B, S), float("nan"), device=DEV)
            hip.seanet_uptail(dev(x16), Wd, bd, w1d, b1d, w2d, b2d, wfd, bf_, wav16, B=B, T=T, x_seg_stride=xs, wav_seg_stride=S)
            torch.cuda.synchronize()
            outs.append(wav16.cpu())
    finally:
        hip.load().sopro_seanet_uptail_set_tiles(0)
    assert torch.equal(outs[0], outs[1])
    assert torch.equal(outs[0], wav.cpu())  # the same rounded operands either way
    assert float((outs[0] - full).abs().max()) <= 3e-2 * peak, _report(outs[0], full, "bf16 rows vs fp32 layers")
    with pytest.raises(hip.SoproHipError):
        hip.seanet_uptail(dev(xb), Wd, bd, w1d, b1d, w2d, b2d, wfd, bf_, wav, B=B, T=T, x_seg_stride=xs, wav_seg_stride=S, passes=2)
