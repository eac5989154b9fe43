"""The C-side checkpoint loader (csrc/checkpoint.hip: sopro_checkpoint_open / _tensor / _engine_cfg) on the CPU: a host that is not
Python starts from the reference's own files - model.safetensors with SoproTTSModel.state_dict() keys and its config JSON in the
header (src/sopro/hub.py:30-52), the Mimi codec's safetensors with HuggingFace MimiModel.state_dict() keys - and must arrive at the
SAME packed operands the Python host makes with sopro_amd/pack.py.  No GPU needed (tests/test_gpu_stages.py runs an utterance on an
engine built this way)."""
import ctypes as C
import os

import numpy as np
import pytest
import torch

from sopro_amd import hip
from sopro_amd.config import MimiDecoderConfig, SoproTTSConfig
from sopro_amd.pack import pack_mimi, pack_sopro, rope_tables, sinusoid_table
from sopro_amd.weights import save_sopro_checkpoint, synth_mimi_weights, synth_sopro_weights


def _open(tmp_path, cfg, wn, mn, dtype=None):
    from safetensors.numpy import save_file

    sp, mp = str(tmp_path / "model.safetensors"), str(tmp_path / "mimi.safetensors")
    save_sopro_checkpoint(sp, wn, cfg)
    if dtype is not None:  # a half-precision checkpoint (the loader reads F16 / BF16 / F64 as fp32)
        from safetensors.torch import save_file as save_pt

        save_pt({k: torch.from_numpy(np.require(v, requirements="C")).to(dtype) for k, v in wn.items()}, sp, metadata={"cfg": cfg.to_json()})
    save_file({k: np.require(v, requirements="C") for k, v in mn.items()}, mp)
    lib = hip.load()
    ck = C.c_void_p()
    hip._check(lib.sopro_checkpoint_open(sp.encode(), mp.encode(), C.byref(ck)), "sopro_checkpoint_open")
    out = {}
    for i in range(lib.sopro_checkpoint_count(ck)):
        name, data, shape, nd = C.c_char_p(), C.c_void_p(), (C.c_int64 * 4)(), C.c_int32()
        hip._check(lib.sopro_checkpoint_tensor(ck, i, C.byref(name), C.byref(data), shape, C.byref(nd)), "sopro_checkpoint_tensor")
        shp = tuple(int(shape[d]) for d in range(nd.value))
        n = int(np.prod(shp)) if shp else 1
        out[name.value.decode()] = np.ctypeslib.as_array(C.cast(data, C.POINTER(C.c_float)), shape=(n,)).reshape(shp).copy()
    return lib, ck, out


def test_c_loader_packs_what_pack_py_packs(tmp_path):
    cfg, mc = SoproTTSConfig(), MimiDecoderConfig()
    wn, mn = synth_sopro_weights(cfg, 512, 11), synth_mimi_weights(mc, 11, with_encoder=True)
    lib, ck, got = _open(tmp_path, cfg, wn, mn)
    try:
        want = dict(pack_sopro(wn, cfg))
        want.update(pack_mimi(mn, mc))
        want["pe"] = sinusoid_table(int(cfg.pos_emb_max) + 8, int(cfg.d_model))
        want["rope.cos"], want["rope.sin"] = rope_tables(8192, int(mc.head_dim), float(mc.rope_theta))
        missing = [k for k in want if k not in got and want[k].dim() >= 1]
        assert not missing, missing[:8]
        exact = inexact = 0
        for k, w in want.items():
            if w.dim() < 1:
                continue
            g, w = torch.from_numpy(got[k]), w.float()
            assert tuple(g.shape) == tuple(w.shape), (k, tuple(g.shape), tuple(w.shape))
            if torch.equal(g, w):
                exact += 1
                continue
            inexact += 1
            if k in ("pe", "rope.cos", "rope.sin"):
                # torch's vectorised expf / powf (1-ulp class) and libm's differ in the last bit of a few of the 192 / 32 frequencies; a
                # frequency that differs moves its whole column pair by up to an ulp of the ARGUMENT (positions reach 4103 / 8191):
                # all but a handful of columns must agree to the last bit of the result, the rest to an ulp of their argument
                col = (g - w).abs().amax(dim=0)
                assert int((col > 2e-7).sum()) <= 6 and float(col.max()) <= 1e-3, (k, int((col > 2e-7).sum()), float(col.max()))
                assert float((g[:512] - w[:512]).abs().max()) <= 4e-5  # ... and the rows a 400-frame utterance can reach
                continue
            # float64 folds / softmax / tanh / table entries: the same value up to the last bit of an fp32 result
            # (the tables' sin / cos arguments reach 4104 * 1.0: an ulp of the fp32 argument is 5e-4 there, both sides take the same one)
            assert k.endswith((".qu.w", ".q.b", ".qa.w", ".gate_scale", ".b", "enc.cb_bias")) or k.startswith(("nar.mix.", "pe", "rope.", "token2sv.cw", "ref_cw")), k
            tol = 2e-6 * float(w.abs().max()) + 1e-7
            assert float((g - w).abs().max()) <= tol, (k, float((g - w).abs().max()), tol)
        assert exact > 200 and inexact < 40, (exact, inexact)
        # the engine configuration a C host derives from the header's cfg JSON
        ecfg = hip.EngineCfg()
        hip._check(lib.sopro_checkpoint_engine_cfg(ck, 0, C.byref(ecfg)), "sopro_checkpoint_engine_cfg")
        assert (ecfg.d_model, ecfg.codebook_size, ecfg.num_codebooks, ecfg.bos_row) == (384, 2048, 32, 65536)
        assert list(ecfg.ar_dilations)[:6] == list(cfg.ar_dilations) and [i for i in range(6) if ecfg.ar_xattn[i]] == list(cfg.ar_xattn_layers)
        for i in cfg.ar_xattn_layers:
            assert abs(ecfg.ar_gate[i] - float(want[f"ar.x_attns.{i}.gate_scale"][0])) < 1e-6
        assert ecfg.n_stages == 4 and list(ecfg.stage_first_cb)[:4] == [1, 4, 8, 16] and list(ecfg.stage_n_cb)[:4] == [3, 4, 8, 16]
        for i, s in enumerate(cfg.stage_order()):
            assert abs(ecfg.nar_mix[i][0] - float(want[f"nar.mix.{s}"][0])) < 1e-6
        assert list(ecfg.mimi_ratios)[:4] == [8, 6, 5, 4] and ecfg.mimi_rope_positions == 8192 and ecfg.mimi_window == 250
        assert abs(ecfg.mimi_final_bias - float(want["sea.final.b"][0])) == 0.0 and ecfg.enc_kernel == 7 and ecfg.precision == 0
    finally:
        lib.sopro_checkpoint_close(ck)


def test_c_loader_reads_a_bf16_checkpoint_and_a_changed_config(tmp_path):
    """Half-precision tensors come back as fp32 (the values the reference's load_state_dict would copy into fp32 parameters), and the
    header's config wins over the defaults (key intersection, src/sopro/hub.py:44-48; unknown keys ignored)."""
    cfg, mc = SoproTTSConfig(n_layers_nar=4, ar_dilation_cycle=(1, 3), stage_B=(2, 3)), MimiDecoderConfig()
    wn, mn = synth_sopro_weights(cfg, 64, 5), synth_mimi_weights(mc, 5)
    lib, ck, got = _open(tmp_path, cfg, wn, mn, dtype=torch.bfloat16)
    try:
        w16 = {k: torch.from_numpy(v).to(torch.bfloat16).float().numpy() for k, v in wn.items()}
        want = pack_sopro(w16, cfg)
        assert torch.equal(torch.from_numpy(got["nar.blocks.3.glu.w"]), want["nar.blocks.3.glu.w"]) and "nar.blocks.4.glu.w" not in got
        assert torch.equal(torch.from_numpy(got["ar.blocks.2.ff1.w"]), want["ar.blocks.2.ff1.w"])
        ecfg = hip.EngineCfg()
        hip._check(lib.sopro_checkpoint_engine_cfg(ck, 1, C.byref(ecfg)), "sopro_checkpoint_engine_cfg")
        assert ecfg.n_layers_nar == 4 and list(ecfg.ar_dilations)[:6] == [1, 3, 1, 3, 1, 3] and ecfg.precision == 1
        assert list(ecfg.stage_first_cb)[:2] == [1, 4] and list(ecfg.stage_n_cb)[:2] == [2, 4]
    finally:
        lib.sopro_checkpoint_close(ck)


def test_c_loader_error_paths(tmp_path):
    lib = hip.load()
    ck = C.c_void_p()
    assert lib.sopro_checkpoint_open(str(tmp_path / "nope.safetensors").encode(), None, C.byref(ck)) != 0
    assert b"cannot open" in lib.sopro_last_error()
    from safetensors.numpy import save_file

    p = str(tmp_path / "nocfg.safetensors")
    save_file({"a": np.zeros(3, dtype=np.float32)}, p)
    assert lib.sopro_checkpoint_open(p.encode(), None, C.byref(ck)) != 0  # src/sopro/hub.py:37-39
    assert b"cfg" in lib.sopro_last_error()
    cfg = SoproTTSConfig()
    wn = synth_sopro_weights(cfg, 64, 1)
    del wn["ar.head.bias"]
    p2 = str(tmp_path / "missing.safetensors")
    save_sopro_checkpoint(p2, wn, cfg)
    assert lib.sopro_checkpoint_open(p2.encode(), None, C.byref(ck)) != 0
    assert b"ar.head.bias" in lib.sopro_last_error()


def _write_raw_safetensors(path, header: bytes, data: bytes = b""):
    import struct

    with open(path, "wb") as f:
        f.write(struct.pack("<Q", len(header)) + header + data)


def _open_rc(lib, sp, mp=None):
    ck = C.c_void_p()
    rc = lib.sopro_checkpoint_open(sp.encode(), mp.encode() if mp else None, C.byref(ck))
    if rc == 0:
        lib.sopro_checkpoint_close(ck)
    return rc, lib.sopro_last_error().decode(errors="replace")


def test_c_loader_refuses_tensors_that_do_not_fit_the_config(tmp_path):
    """the loader indexes tensors with extents taken from the cfg JSON, so every tensor's rank and shape is
    checked against the config BEFORE a pack / fold loop runs, config integers are range-checked, and no exception leaves the C
    entry points.  Each case used to corrupt the heap, terminate the process or return rc 0 with garbage."""
    lib = hip.load()
    cfg = SoproTTSConfig()
    wn = synth_sopro_weights(cfg, 64, 3)
    # (a) cfg d_model = 1536 over 384-wide tensors
    big = SoproTTSConfig(d_model=1536)
    p = str(tmp_path / "a.safetensors")
    save_sopro_checkpoint(p, wn, big)
    rc, err = _open_rc(lib, p)
    assert rc != 0 and "shape" in err and "config implies" in err, err
    # (b) a tensor of the wrong rank (1-D depthwise weight)
    w2 = dict(wn)
    w2["ar.blocks.0.dw.dw.weight"] = wn["ar.blocks.0.dw.dw.weight"].reshape(-1)
    p = str(tmp_path / "b.safetensors")
    save_sopro_checkpoint(p, w2, cfg)
    rc, err = _open_rc(lib, p)
    assert rc != 0 and "ar.blocks.0.dw.dw.weight" in err, err
    # (c) config integers out of range / not integers at all
    import json

    from safetensors.numpy import save_file

    for bad in ({"pos_emb_max": -5}, {"d_model": 1e300}, {"n_layers_ar": 400}, {"num_codebooks": 0}, {"ar_dilation_cycle": [1, -2]},
                {"stage_B": [5, 99999]}, {"ref_xattn_heads": 7}, {"d_model": "384"}, {"nar_head_dim": 2.5}):
        d = json.loads(cfg.to_json())
        d.update(bad)
        p = str(tmp_path / "c.safetensors")
        save_file({k: np.require(v, requirements="C") for k, v in wn.items()}, p, metadata={"cfg": json.dumps(d)})
        rc, err = _open_rc(lib, p)
        assert rc != 0 and ("config" in err or "cfg" in err), (bad, err)
    # (d) a norm vector shorter than the K it scales
    w3 = dict(wn)
    w3["ar.blocks.1.norm.weight"] = wn["ar.blocks.1.norm.weight"][:100].copy()
    p = str(tmp_path / "d.safetensors")
    save_sopro_checkpoint(p, w3, cfg)
    rc, err = _open_rc(lib, p)
    assert rc != 0 and "ar.blocks.1.norm.weight" in err, err
    # Mimi side: a transposed-convolution weight with swapped channel extents
    mc = MimiDecoderConfig()
    mn = synth_mimi_weights(mc, 3)
    mn["decoder.layers.2.conv.weight"] = np.ascontiguousarray(mn["decoder.layers.2.conv.weight"].transpose(1, 0, 2))
    sp, mp = str(tmp_path / "ok.safetensors"), str(tmp_path / "mimi_bad.safetensors")
    save_sopro_checkpoint(sp, wn, cfg)
    save_file({k: np.require(v, requirements="C") for k, v in mn.items()}, mp)
    rc, err = _open_rc(lib, sp, mp)
    assert rc != 0 and "decoder.layers.2.conv.weight" in err, err


def test_c_loader_survives_hostile_safetensors_headers(tmp_path):
    """offsets / dims are checked non-negative integers, byte counts must equal numel * element size before anything
    is allocated, numbers are parsed from a NUL-terminated copy of the header, the JSON depth is capped, \\u escapes become UTF-8."""
    import json

    lib = hip.load()
    cases = {
        "neg_offset": (json.dumps({"a": {"dtype": "F32", "shape": [2], "data_offsets": [-8, 0]}}).encode(), b"\0" * 8),
        "nan_dim": (b'{"a":{"dtype":"F32","shape":[NaN],"data_offsets":[0,8]}}', b"\0" * 8),
        "neg_dim": (json.dumps({"a": {"dtype": "F32", "shape": [-2], "data_offsets": [0, 8]}}).encode(), b"\0" * 8),
        "huge_dim": (json.dumps({"a": {"dtype": "F32", "shape": [2 ** 40, 2 ** 40], "data_offsets": [0, 8]}}).encode(), b"\0" * 8),
        "huge_off": (b'{"a":{"dtype":"F32","shape":[2],"data_offsets":[0,1e300]}}', b"\0" * 8),
        "wrap_off": (json.dumps({"a": {"dtype": "F32", "shape": [2], "data_offsets": [0, 2 ** 64 - 4]}}).encode(), b"\0" * 8),
        "bytes_vs_numel": (json.dumps({"a": {"dtype": "F32", "shape": [1000000], "data_offsets": [0, 8]}}).encode(), b"\0" * 8),
        "frac_dim": (b'{"a":{"dtype":"F32","shape":[1.5],"data_offsets":[0,8]}}', b"\0" * 8),
        "deep": (b"[" * 100000, b""),
        "deep_obj": (b'{"a":' * 50000, b""),
        "digits_at_end": (b'{"a":{"dtype":"F32","shape":[2],"data_offsets":[0,8', b"99999999"),  # the number runs into the tensor data
        "shape_not_list": (json.dumps({"a": {"dtype": "F32", "shape": 2, "data_offsets": [0, 8]}}).encode(), b"\0" * 8),
        "dtype_not_str": (json.dumps({"a": {"dtype": 7, "shape": [2], "data_offsets": [0, 8]}}).encode(), b"\0" * 8),
        "bad_escape": (b'{"a\\uZZZZ":{"dtype":"F32","shape":[2],"data_offsets":[0,8]}}', b"\0" * 8),
    }
    for name, (hdr, data) in cases.items():
        p = str(tmp_path / f"{name}.safetensors")
        _write_raw_safetensors(p, hdr, data)
        rc, err = _open_rc(lib, p)
        assert rc != 0 and err, name
    # a well-formed header whose cfg string holds escapes: a surrogate pair and a BMP character survive as UTF-8 in an ignored key
    cfg = SoproTTSConfig()
    d = json.loads(cfg.to_json())
    d["note é\U0001F600"] = 1
    hdr = json.dumps({"__metadata__": {"cfg": json.dumps(d)}, "x": {"dtype": "F32", "shape": [2], "data_offsets": [0, 8]}}).encode()
    assert b"\\\\ud83d" in hdr  # (json.dumps escaped the pair inside the nested string)
    p = str(tmp_path / "esc.safetensors")
    _write_raw_safetensors(p, hdr, b"\0" * 8)
    rc, err = _open_rc(lib, p)
    assert rc != 0 and "is missing" in err, err  # the header and the cfg parse; the first tensor the packer asks for is absent


def test_c_loader_accepts_a_disabled_stage(tmp_path):
    """A stage pair with last < first is an EMPTY stage in the reference (_stage_range_to_indices, src/sopro/model.py:39-42; stage_order
    skips it, :92-94): such a checkpoint opens and packs what pack.py packs (it used to be refused)."""
    cfg, mc = SoproTTSConfig(stage_C=(5, 16), stage_D=(9, 8)), MimiDecoderConfig()
    assert cfg.stage_order() == ["B", "C", "E"]
    wn, mn = synth_sopro_weights(cfg, 512, 13), synth_mimi_weights(mc, 13)
    lib, ck, got = _open(tmp_path, cfg, wn, mn)
    try:
        want = pack_sopro(wn, cfg)
        for k, w in want.items():
            if k.startswith("nar.") and w.dim() >= 1:
                assert k in got and tuple(got[k].shape) == tuple(w.shape), k
                assert torch.allclose(torch.from_numpy(got[k]), w.float(), rtol=1e-6, atol=1e-7), k
    finally:
        lib.sopro_checkpoint_close(ck)
