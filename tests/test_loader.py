"""Checkpoint / IO formats of SURVEY.md 8f-4: ``model.safetensors`` with the config in ``__metadata__["cfg"]``
(reference: src/sopro/hub.py:30-52), dtype handling, and ``SoproTTS.from_pretrained(local_dir)``
(reference: src/sopro/model.py:419-451).  CPU parts run everywhere; building the engine needs the GPU."""
import json
import os
import sys

import numpy as np
import pytest
import torch

from sopro_amd.config import MimiDecoderConfig, SoproTTSConfig
from sopro_amd.weights import (load_cfg_from_safetensors, load_safetensors, save_sopro_checkpoint, sopro_weight_spec,
                               check_against_spec, synth_mimi_weights, synth_sopro_weights)

VOCAB = 64


def test_checkpoint_round_trip_keeps_config_and_tensors(tmp_path):
    cfg = SoproTTSConfig(style_strength=1.3, min_gen_frames=17)
    w = synth_sopro_weights(cfg, VOCAB, 3)
    p = str(tmp_path / "model.safetensors")
    save_sopro_checkpoint(p, w, cfg)
    cfg2 = load_cfg_from_safetensors(p)
    assert cfg2 == cfg and isinstance(cfg2.ar_dilation_cycle, tuple) and isinstance(cfg2.stage_B, tuple)
    w2 = load_safetensors(p)
    assert set(w2) == set(w) and not check_against_spec(w2, sopro_weight_spec(cfg, VOCAB), what="round trip")
    assert all(np.array_equal(w[k], w2[k]) for k in w)
    sub = load_safetensors(p, names=["ar.head.weight", "not.there"])
    assert list(sub) == ["ar.head.weight"]


def test_cfg_metadata_is_a_key_intersection_like_the_reference(tmp_path):
    """hub.py:44-48: unknown keys are dropped, missing keys keep their defaults, lists become tuples."""
    from safetensors.torch import save_file

    p = str(tmp_path / "m.safetensors")
    save_file({"x": torch.zeros(2)}, p, metadata={"cfg": json.dumps({"d_model": 384, "ar_dilation_cycle": [1, 2, 4], "from_the_future": 9})})
    cfg = load_cfg_from_safetensors(p)
    assert cfg.ar_dilation_cycle == (1, 2, 4) and cfg.num_codebooks == SoproTTSConfig().num_codebooks and not hasattr(cfg, "from_the_future")
    q = str(tmp_path / "nocfg.safetensors")
    save_file({"x": torch.zeros(2)}, q)
    with pytest.raises(RuntimeError, match="No 'cfg' metadata"):
        load_cfg_from_safetensors(q)


@pytest.mark.skipif(not os.path.isdir("/root/reference/src/sopro"), reason="the reference checkout exists in the build container only")
def test_the_reference_reader_accepts_our_checkpoint(tmp_path):
    """Our writer -> the reference's own load_cfg_from_safetensors / load_state_dict_from_safetensors (hub.py:30-52)."""
    sys.dont_write_bytecode = True
    sys.path.insert(0, "/root/reference/src")
    try:
        from sopro.hub import load_cfg_from_safetensors as ref_cfg, load_state_dict_from_safetensors as ref_sd
    finally:
        sys.path.remove("/root/reference/src")
    cfg = SoproTTSConfig(style_strength=1.25)
    w = synth_sopro_weights(cfg, VOCAB, 4)
    p = str(tmp_path / "model.safetensors")
    save_sopro_checkpoint(p, w, cfg)
    rc = ref_cfg(p)
    assert float(rc.style_strength) == 1.25 and tuple(rc.ar_dilation_cycle) == tuple(cfg.ar_dilation_cycle)
    sd = ref_sd(p)
    assert set(sd) == set(w) and torch.equal(sd["ar.head.weight"], torch.from_numpy(w["ar.head.weight"]))


def test_half_precision_checkpoints_load_as_float32(tmp_path):
    """bf16 / fp16 tensors (numpy has no bf16) come back as float32; integer tensors keep their type."""
    from safetensors.torch import save_file

    x = torch.randn(5, 7)
    p = str(tmp_path / "h.safetensors")
    save_file({"a": x.to(torch.bfloat16), "b": x.to(torch.float16), "i": torch.arange(4, dtype=torch.int64)}, p)
    out = load_safetensors(p)
    assert out["a"].dtype == np.float32 and out["b"].dtype == np.float32 and out["i"].dtype == np.int64
    assert np.array_equal(out["a"], x.to(torch.bfloat16).float().numpy())


def _write_tokenizer(d):
    from tokenizers import Tokenizer, models, pre_tokenizers
    from transformers import PreTrainedTokenizerFast

    vocab = {"<s>": 0, "</s>": 1, "<unk>": 2, **{w: i + 3 for i, w in enumerate("hello world this is sopro on mi355x".split())}}
    tk = Tokenizer(models.WordLevel(vocab, unk_token="<unk>"))
    tk.pre_tokenizer = pre_tokenizers.Whitespace()
    PreTrainedTokenizerFast(tokenizer_object=tk, bos_token="<s>", eos_token="</s>", unk_token="<unk>").save_pretrained(d)
    return vocab


def test_tokenizer_wrapper_adds_bos_eos_like_the_reference(tmp_path):
    """src/sopro/tokenizer.py:15-38: a pad token is added when missing, ids are wrapped in BOS / EOS."""
    from sopro_amd.tts import _load_tokenizer

    vocab = _write_tokenizer(str(tmp_path))
    tok = _load_tokenizer(str(tmp_path))
    assert tok.encode("hello sopro") == [0, vocab["hello"], vocab["sopro"], 1]
    assert tok.tok.pad_token == "<|pad|>" and tok.vocab_size == tok.tok.vocab_size + len(tok.tok.get_added_vocab()) > len(vocab)


def test_from_pretrained_without_a_checkpoint_raises_like_the_reference(tmp_path):
    from sopro_amd import SoproTTS

    with pytest.raises(FileNotFoundError, match="model.safetensors"):  # model.py:437-438
        SoproTTS.from_pretrained(str(tmp_path), device="cuda:0")


@pytest.mark.gpu
def test_from_pretrained_local_dir_matches_from_weights(tmp_path):
    """A snapshot directory (model.safetensors with cfg metadata, tokenizer files, mimi/model.safetensors) -> an engine
    that synthesizes the same samples as one built from the in-memory arrays."""
    from safetensors.numpy import save_file
    from sopro_amd import SoproTTS
    from sopro_amd.tts import _load_tokenizer

    d = str(tmp_path)
    _write_tokenizer(d)
    tok = _load_tokenizer(d)
    cfg, mc = SoproTTSConfig(style_strength=1.1), MimiDecoderConfig()
    w, mw = synth_sopro_weights(cfg, tok.vocab_size, 5), synth_mimi_weights(mc, 5)
    save_sopro_checkpoint(os.path.join(d, "model.safetensors"), w, cfg)
    os.makedirs(os.path.join(d, "mimi"))
    save_file({k: np.require(v, requirements="C") for k, v in mw.items()}, os.path.join(d, "mimi", "model.safetensors"))
    a = SoproTTS.from_pretrained(d, device="cuda:0")
    assert a.cfg == cfg and a.device == torch.device("cuda:0")
    b = SoproTTS.from_weights(cfg, w, mw, tok, device="cuda:0")
    ref_tq = torch.from_numpy(np.random.default_rng(5).integers(0, 2048, size=(20, 32)))
    kw = dict(ref_tokens_tq=ref_tq, max_frames=8, top_p=0.0, temperature=1.0, anti_loop=False)
    wa, wb = a.synthesize("hello world this is sopro", **kw), b.synthesize("hello world this is sopro", **kw)
    assert wa.shape == wb.shape and wa.shape[-1] > 0 and torch.equal(wa, wb)


def test_export_for_non_python_hosts(tmp_path):
    """python -m sopro_amd.export: every tensor the stage-level C entry points ask for (sopro_engine_finalize's list) is in the
    flat file, 256-byte aligned, and reads back as the packed array."""
    from sopro_amd.export import export_packed
    from sopro_amd.pack import pack_mimi, pack_sopro

    cfg, mc = SoproTTSConfig(), MimiDecoderConfig()
    w, mw = synth_sopro_weights(cfg, VOCAB, 6), synth_mimi_weights(mc, 6)
    meta = export_packed(w, mw, cfg, str(tmp_path / "packed"), rope_positions=64)
    names = {t["name"]: t for t in meta["tensors"]}
    for need in ("ar.blocks.0.glu.w", "ar.x_attns.1.q.wT", "ar.head.w", "cb_embed", "nar.blocks.5.ff2.w", "nar.heads.B.w", "nar.heads.E.b",
                 "nar.adapter.mlp.2.w", "codebooks", "rvq_proj.w", "tr.7.fc2.w", "sea.conv0.w", "sea.up3.w", "sea.res2.c1.w", "sea.final.w",
                 "rope.cos", "upsample.w", "pe", "text_enc.embed", "text_enc.layers.0.glu.w", "ref_enc_blocks.0.ff2.w", "ref_xattn.blocks.2.kv.w",
                 "spk_film.mlp.0.w", "token2sv.proj.w", "cond_norm.weight"):
        assert need in names, need
    blob = np.fromfile(str(tmp_path / "packed.bin"), dtype=np.uint8)
    assert blob.size == meta["bytes"] and all(t["offset"] % 256 == 0 for t in meta["tensors"])
    ps, pm = pack_sopro(w, cfg), pack_mimi(mw, mc)
    for name, src in (("ar.head.w", ps), ("sea.up1.w", pm), ("nar.heads.C.w", None)):
        t = names[name]
        a = np.frombuffer(blob, dtype=np.float32, count=int(np.prod(t["shape"])), offset=t["offset"]).reshape(t["shape"])
        want = (src[name] if src is not None else ps["nar.heads.C.w"]).numpy()
        assert np.array_equal(a, want), name
    c = meta["cfg"]
    assert c["n_layers_text"] == cfg.n_layers_text and c["ref_xattn_layers"] == cfg.ref_xattn_layers and c["enc_kernel"] == 7
    assert c["stage_first_cb"] == [1, 4, 8, 16] and c["stage_n_cb"] == [3, 4, 8, 16] and c["ar_xattn"] == [0, 1, 0, 1, 0, 1] and c["mimi_ratios"] == [8, 6, 5, 4]
