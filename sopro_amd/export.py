"""Write the repacked weights of a checkpoint as one flat file for hosts that do not run Python:

    python -m sopro_amd.export <snapshot_dir with model.safetensors (+ mimi/model.safetensors)> <out_prefix>

-> ``<out_prefix>.bin`` (little-endian tensors back to back, 256-byte aligned) and ``<out_prefix>.json``
(``{"cfg": {sopro_engine_cfg fields}, "tensors": [{"name", "dtype", "shape", "offset"}]}``).  A C host mmaps the file,
copies it to the device once and calls ``sopro_engine_set_tensor(engine, name, base + offset, shape, ndim)`` per entry
(INTEGRATION.md).  The names are the keys of ``sopro_amd.pack.pack_sopro`` / ``pack_mimi`` (all host-side repacking -
GLU interleave, tap-major convolution weights, folded norm vectors and head-id embeddings - is already applied), plus
``rope.cos`` / ``rope.sin`` and the positional aliases ``nar.heads.<B|C|D|E by position>``.  CPU only: no GPU needed."""
from __future__ import annotations

import json
import math
import os
import sys
from typing import Dict

import numpy as np
import torch

from .config import MimiDecoderConfig
from .hip import ABI_VERSION  # the constant only: nothing is loaded
from .pack import pack_mimi, pack_sopro, rope_tables
from .weights import load_cfg_from_safetensors, load_safetensors

_POS = "BCDEFGHI"


def export_packed(weights: Dict[str, np.ndarray], mimi_weights: Dict[str, np.ndarray], cfg, out_prefix: str, rope_positions: int = 8192) -> Dict:
    mc = MimiDecoderConfig(num_quantizers=int(cfg.num_codebooks))
    ps, pm = pack_sopro(weights, cfg), pack_mimi(mimi_weights, mc)
    cos, sin = rope_tables(rope_positions, int(mc.head_dim), float(mc.rope_theta))
    tensors = dict(ps)
    tensors.update(pm)
    tensors["rope.cos"], tensors["rope.sin"] = cos, sin
    from .pack import sinusoid_table

    tensors["pe"] = sinusoid_table(int(cfg.pos_emb_max) + 8, int(cfg.d_model))  # position table of the conditioning family (src/sopro/model.py:62-64)
    order, sc = cfg.stage_order(), cfg.stage_codebooks()
    if len(order) > len(_POS):
        raise ValueError(f"{len(order)} NAR stages, the stage-level C API names at most {len(_POS)}")
    for i, s in enumerate(order):
        tensors[f"nar.heads.{_POS[i]}.w"], tensors[f"nar.heads.{_POS[i]}.b"] = ps[f"nar.heads.{s}.w"], ps[f"nar.heads.{s}.b"]
    ecfg = {
        "d_model": int(cfg.d_model), "codebook_size": int(cfg.codebook_size), "num_codebooks": int(cfg.num_codebooks), "nar_head_dim": int(cfg.nar_head_dim),
        "bos_row": int(cfg.bos_row), "n_layers_ar": int(cfg.n_layers_ar), "ar_kernel": int(cfg.ar_kernel), "ar_dilations": [int(d) for d in cfg.ar_dilations],
        "ar_xattn": [int(i in cfg.ar_xattn_layers) for i in range(int(cfg.n_layers_ar))],
        "ar_gate": [float(ps[f"ar.x_attns.{i}.gate_scale"][0]) if i in cfg.ar_xattn_layers else 0.0 for i in range(int(cfg.n_layers_ar))],
        "n_layers_nar": int(cfg.n_layers_nar), "nar_kernel": int(cfg.nar_kernel_size), "nar_dilations": [int(d) for d in cfg.nar_dilations],
        "n_stages": len(order), "stage_first_cb": [int(sc[s][0]) for s in order], "stage_n_cb": [len(sc[s]) for s in order],
        "nar_mix": [[float(ps[f"nar.mix.{s}"][0]), float(ps[f"nar.mix.{s}"][1])] for s in order],
        "nar_prev_cb_weights": [float(v) for v in ps["nar_prev_cb_weights"].float()],
        "mimi_hidden": int(mc.hidden_size), "mimi_codebook_dim": int(mc.codebook_dim), "mimi_heads": int(mc.num_attention_heads),
        "mimi_head_dim": int(mc.head_dim), "mimi_layers": int(mc.num_hidden_layers), "mimi_window": int(mc.sliding_window),
        "mimi_inter": int(mc.intermediate_size), "mimi_n_ratios": len(mc.upsampling_ratios), "mimi_ratios": [int(r) for r in mc.upsampling_ratios],
        "mimi_num_filters": int(mc.num_filters), "mimi_kernel": int(mc.kernel_size), "mimi_res_kernel": int(mc.residual_kernel_size),
        "mimi_last_kernel": int(mc.last_kernel_size), "mimi_compress": int(mc.compress), "mimi_n_semantic": int(mc.num_semantic_quantizers),
        "mimi_rope_positions": int(rope_positions), "mimi_norm_eps": float(mc.norm_eps), "mimi_final_bias": float(pm["sea.final.b"][0]), "precision": 0,
        "n_layers_text": int(cfg.n_layers_text), "ref_enc_layers": int(cfg.ref_enc_layers), "ref_xattn_layers": int(cfg.ref_xattn_layers),
        "ref_xattn_heads": int(cfg.ref_xattn_heads), "sv_student_dim": int(cfg.sv_student_dim), "enc_kernel": 7,
    }
    table, off = [], 0
    with open(out_prefix + ".bin", "wb") as f:
        for name in sorted(tensors):
            t = tensors[name]
            if t.dim() > 4:
                raise ValueError(f"tensor {name!r} has {t.dim()} dimensions, the stage-level C API takes 1..4")
            if t.dim() < 1:
                t = t.reshape(1)  # scalars travel as one-element vectors (sopro_engine_set_tensor wants ndim >= 1)
            a = (t.to(torch.float32) if t.is_floating_point() else t.to(torch.int32)).contiguous().numpy()
            pad = (-off) % 256
            f.write(b"\0" * pad)
            off += pad
            table.append({"name": name, "dtype": "f32" if a.dtype == np.float32 else "i32", "shape": list(a.shape), "offset": off})
            f.write(a.tobytes(order="C"))
            off += a.nbytes
    meta = {"cfg": ecfg, "tensors": table, "bytes": off, "abi_version": ABI_VERSION}
    with open(out_prefix + ".json", "w") as f:
        json.dump(meta, f)
    return meta


def main(argv) -> int:
    if len(argv) != 3:
        print(__doc__)
        return 2
    snap, out = argv[1], argv[2]
    path = os.path.join(snap, "model.safetensors")
    cfg = load_cfg_from_safetensors(path)
    meta = export_packed(load_safetensors(path), load_safetensors(os.path.join(snap, "mimi", "model.safetensors")), cfg, out)
    print(f"{len(meta['tensors'])} tensors, {meta['bytes'] / 2**20:.1f} MiB -> {out}.bin / {out}.json")
    return 0


if __name__ == "__main__":
    sys.exit(main(sys.argv))
