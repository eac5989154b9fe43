"""Reference-audio path on the GPU: WAV -> trim -> resample -> Mimi encoder -> codes, through the
C-ABI, against the oracle and the HF ``MimiModel.encode`` fixture (tests/golden/mimi_encode.npz)."""
import wave

import numpy as np
import pytest
import torch

from conftest import SEED, golden
from oracle import sopro_oracle as O

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def enc_np(mc):
    from sopro_amd.weights import synth_mimi_weights
    return synth_mimi_weights(mc, SEED, with_encoder=True)


@pytest.fixture(scope="module")
def codec(mc, enc_np):
    from sopro_amd.codec import MimiCodec
    return MimiCodec(enc_np, mc, device="cuda:0")


@pytest.fixture(scope="module")
def mwe(enc_np):
    return O.to_torch(enc_np)


def _check_codes(got_tq, wav_n, mwe, mc, want_tq=None, min_exact=0.9):
    """Codes must equal the oracle's; where a frame departs, the first differing layer must be a numerical tie
    (the two candidate codes are equidistant from the residual to within fp32 round-off)."""
    taps = {}
    want = O.mimi_encode(wav_n.view(1, 1, -1), mwe, mc, taps)[0].permute(1, 0) if want_tq is None else want_tq
    got = got_tq.cpu()
    assert got.shape == want.shape
    cbs = O.mimi_codebooks(mwe, mc)
    ns = int(mc.num_semantic_quantizers)
    ds = taps["enc_downsample"] if taps else None
    exact = 0
    for t in range(want.shape[0]):
        if torch.equal(got[t], want[t]):
            exact += 1
            continue
        if ds is None:
            continue
        q = int((got[t] != want[t]).nonzero()[0, 0])
        grp, q0 = ("semantic", 0) if q < ns else ("acoustic", ns)
        res = torch.nn.functional.conv1d(ds[:, :, t:t + 1], mwe[f"quantizer.{grp}_residual_vector_quantizer.input_proj.weight"])[0, :, 0]
        for qq in range(q0, q):
            res = res - cbs[qq][want[t, qq]]
        da = float((res - cbs[q][want[t, q]]).norm())
        db = float((res - cbs[q][got[t, q]]).norm())
        assert abs(da - db) <= 2e-4 * max(da, 1e-6), f"frame {t} layer {q}: not a tie ({da} vs {db})"
    assert exact >= min_exact * want.shape[0], f"only {exact}/{want.shape[0]} frames identical"


def test_fir1_matches_conv1d():
    from sopro_amd import hip

    g = torch.Generator().manual_seed(3)
    for (B, n, C, K, stride, left) in ((2, 1000, 64, 7, 1, 6), (1, 2999, 80, 171, 147, 12), (1, 50, 3, 17, 2, 8)):
        x = torch.randn(B, n, generator=g)
        w = torch.randn(C, K, generator=g)
        b = torch.randn(C, generator=g)
        n_out = (n + left) // stride + 1
        xp = torch.nn.functional.pad(x[:, None], (left, K + stride))
        ref = torch.nn.functional.conv1d(xp.double(), w[:, None].double(), b.double(), stride=stride)[:, :, :n_out].transpose(1, 2)
        out = torch.zeros(B, n_out, C, device="cuda:0")
        hip.fir1(x.cuda(), w.cuda(), out, B=B, n_in=n, n_out=n_out, C_=C, K=K, stride=stride, left=left, bias=b.cuda())
        torch.cuda.synchronize()
        assert torch.allclose(out.cpu().double(), ref, atol=2e-4 * (K ** 0.5))


def test_rvq_assign_matches_argmin_and_updates_residual():
    from sopro_amd import hip

    g = torch.Generator().manual_seed(4)
    rows, V, D = 37, 2048, 256
    tab = torch.randn(3 * V, D, generator=g)
    res = torch.randn(rows, D, generator=g)
    e = tab[V:2 * V]
    scores = res @ e.t() - 0.5 * (e ** 2).sum(1)
    want = torch.cdist(res[None], e[None])[0].argmin(-1)
    codes = torch.full((rows, 5), -1, dtype=torch.int32, device="cuda:0")
    rd = res.cuda()
    hip.rvq_assign(scores.cuda(), tab.cuda(), rd, codes, rows=rows, V=V, D=D, ldc=5, t_off=V * D, c_off=2)
    torch.cuda.synchronize()
    assert torch.equal(codes[:, 2].cpu().long(), want)
    assert torch.equal(codes[:, 0].cpu(), torch.full((rows,), -1, dtype=torch.int32))
    assert torch.allclose(rd.cpu(), res - e[want], atol=1e-6)
    # exact ties resolve to the first index, like argmin/argmax on the host
    sc = torch.zeros(2, V)
    sc[0, 7] = sc[0, 900] = 3.0
    c2 = torch.zeros(2, 1, dtype=torch.int32, device="cuda:0")
    hip.rvq_assign(sc.cuda(), tab.cuda(), torch.zeros(2, D, device="cuda:0"), c2, rows=2, V=V, D=D, ldc=1)
    assert c2[:, 0].tolist() == [7, 0]


def test_encode_matches_hf_fixture(codec, mc, mwe):
    g = golden("mimi_encode")
    wav = torch.from_numpy(g["wav"])
    got = codec.encode_waveform(wav.cuda())
    assert got.dtype == torch.int64 and got.shape == (13, 32)
    _check_codes(got, wav, mwe, mc)
    assert float((got.cpu() == torch.from_numpy(g["codes"])).float().mean()) > 0.9  # HF MimiModel.encode itself


@pytest.mark.parametrize("n", [1, 700, 1920, 1921, 5 * 1920 + 959, 24000 * 4 + 123])
def test_encode_ragged_lengths(codec, mc, mwe, n):
    g = torch.Generator().manual_seed(n)
    wav = 0.25 * torch.randn(n, generator=g)
    got = codec.encode_waveform(wav.cuda())
    assert got.shape == (-(-n // 1920), 32)
    assert int(got.min()) >= 0 and int(got.max()) < 2048
    _check_codes(got, wav, mwe, mc)


def test_encode_batch_equals_single(codec):
    g = torch.Generator().manual_seed(11)
    wav = 0.25 * torch.randn(3, 9000, generator=g).cuda()
    both = codec.encode_waveform(wav)
    assert both.shape == (3, 5, 32)
    for b in range(3):
        assert torch.equal(both[b], codec.encode_waveform(wav[b]))


def test_encode_then_decode_runs(codec):
    """encode -> decode is the codec's defining round trip; with random weights only shapes/finite-ness are meaningful."""
    g = torch.Generator().manual_seed(12)
    wav = 0.25 * torch.randn(1920 * 6, generator=g).cuda()
    codes = codec.encode_waveform(wav)
    back = codec.decode_full(codes)
    assert back.shape == (1, 1, 6 * 1920) and bool(torch.isfinite(back).all())


@pytest.mark.parametrize("sr", [16000, 44100, 48000, 24000])
def test_resample_matches_oracle(codec, sr):
    g = torch.Generator().manual_seed(sr)
    x = torch.randn(sr // 3 + 17, generator=g)
    got = codec.resample(x.cuda(), sr, 24000).cpu()
    want = O.sinc_resample(x, sr, 24000)
    assert got.shape == want.shape
    assert float((got - want).abs().max()) < 2e-5


def test_encode_file_and_prepare_reference(tmp_path, cfg, mc, sopro_np, enc_np, mwe, w):
    from conftest import FakeTok
    from sopro_amd import SoproTTS

    sr = 16000
    rng = np.random.default_rng(21)
    x = 1e-4 * rng.standard_normal(int(2.9 * sr)).astype(np.float32)
    a, b = int(0.35 * sr), int(2.5 * sr)
    x[a:b] += (0.3 * np.sin(np.arange(b - a) * 0.05) * rng.uniform(0.5, 1.0, b - a)).astype(np.float32)
    path = str(tmp_path / "ref.wav")
    with wave.open(path, "wb") as f:
        f.setnchannels(1), f.setsampwidth(2), f.setframerate(sr)
        f.writeframes(np.round(x * 32767).astype("<i2").tobytes())
    tts = SoproTTS.from_weights(cfg, sopro_np, enc_np, FakeTok(), device="cuda:0")
    xq = torch.from_numpy(np.round(x * 32767).astype(np.int16).astype(np.float32) / 32768.0)
    for crop in (None, 1.0):
        got = tts.codec.encode_file(path, crop_seconds=crop)
        want = O.encode_audio(xq, sr, mwe, mc, crop_seconds=crop)
        assert got.shape == want.shape
        if crop is not None:
            assert got.shape[0] == 12  # round(1.0 s * 12.5 fps) frames
        assert float((got.cpu() == want).float().mean()) > 0.85
    ref = tts.prepare_reference(ref_audio_path=path, ref_seconds=1.5)
    oref = O.prepare_reference(tts.encode_reference(ref_audio_path=path, ref_seconds=1.5).cpu(), w, cfg)
    assert torch.allclose(ref.sv_ref.cpu(), oref.sv_ref, atol=2e-4)
    assert torch.allclose(ref.ref_seq.cpu(), oref.ref_seq, atol=1e-3)
    with pytest.raises(RuntimeError):
        tts.encode_reference(ref_audio_path=path, ref_tokens_tq=torch.zeros(4, 32, dtype=torch.long))
