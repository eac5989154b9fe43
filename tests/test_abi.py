"""The C-ABI library: loads without a GPU, exports every symbol include/sopro_hip.h declares, and
rejects bad arguments with an error code + message instead of launching (no compute calls here)."""
import ctypes
import os
import re

import pytest

from conftest import ROOT
from sopro_amd import hip


def _declared_symbols():
    src = open(os.path.join(ROOT, "include", "sopro_hip.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(sopro_[a-z0-9_]+)\s*\(", src)))


def test_library_loads_and_exports_every_declared_symbol():
    lib = hip.load()
    names = _declared_symbols()
    assert len(names) >= 20
    for n in names:
        assert hasattr(lib, n), f"{n} declared in include/sopro_hip.h but not exported by libsopro_hip.so"
    assert set(names) == set(hip.SYMBOLS), set(names) ^ set(hip.SYMBOLS)
    assert lib.sopro_abi_version() == hip.ABI_VERSION


def test_struct_layouts_match_the_header(tmp_path):
    """Compile include/sopro_hip.h with gcc and compare sizeof/offsetof with the ctypes mirrors:
    field order/size drift between the two would corrupt every launch."""
    import subprocess

    checks = {
        "sopro_gemm_split_ext": (hip.SplitExt, ["a_format", "c_mode", "C2", "ldc2", "c2_seg_stride", "rms_norm", "rms_eps", "ksplit", "n_tickets", "ws", "ws_bytes", "tickets", "group_m"]),
        "sopro_gemm_args": (hip.GemmArgs, ["A", "W", "C", "R", "scale", "pro_vec", "dbg", "M", "rows_per_seg", "epilogue"]),
        "sopro_skinny_args": (hip.SkinnyArgs, ["X", "W", "Y", "scale", "ring", "step", "Xp", "y_part_stride", "dbg", "eps", "B", "epilogue", "ring_len", "ksize", "np", "ksplit", "rms_norm", "w_layout", "aux_W", "aux_Y", "aux_ldr", "aux_y_part_stride", "aux_tiles", "aux_flags", "ring_format", "mt", "nt"]),
        "sopro_attn_args": (hip.AttnArgs, ["Q", "K", "V", "O", "klens", "B", "Tk", "causal", "window", "scale", "kv_index"]),
        "sopro_xattn_args": (hip.XattnArgs, ["X", "Xp", "norm_w", "Kp", "klens", "Y", "eps", "scale", "np", "S_cap", "Qp", "qp_stride", "nqp", "k_unfolded", "kv_format"]),
        "sopro_engine_cfg": (hip.EngineCfg, ["d_model", "bos_row", "ar_dilations", "ar_gate", "nar_dilations", "stage_n_cb", "nar_mix", "nar_prev_cb_weights",
                                             "mimi_hidden", "mimi_ratios", "mimi_compress", "mimi_rope_positions", "mimi_norm_eps", "mimi_final_bias", "precision",
                                             "n_layers_text", "ref_xattn_heads", "sv_student_dim", "enc_kernel"]),
        "sopro_ar_block": (hip.ArBlock, ["glu_w", "dw_b", "ff2_b", "ring", "kp", "vp", "dil", "xattn", "gate", "qa_w", "qu_w", "q_b"]),
        "sopro_ar_frame": (hip.ArFrame, ["blk", "head_w", "x0", "part", "xp", "qa", "qpart", "logits", "klens", "n_layers", "S_cap", "w_layout", "tile_glu", "tile_head", "eps", "k_unfold", "store_format", "st"]),
        "sopro_prof_row": (hip.ProfRow, ["family", "launches", "gpu_bound", "flops", "flops_bound", "ms_all", "ms_bound"]),
        "sopro_mimi_stream_state": (hip.MimiStreamState, ["kv", "cap_rows", "kv_len", "pos", "evict", "half"]),
        "sopro_ar_state": (hip.ArState, ["x_cur", "emb", "hist", "recent", "params", "seed", "B", "bos_row", "start", "row_max", "row_params", "nonce", "dbg"]),
    }
    lines = ["#include <stdio.h>", "#include <stddef.h>", '#include "sopro_hip.h"', "int main(void){"]
    for cname, (_cls, fields) in checks.items():
        lines.append(f'printf("{cname} %zu\\n", sizeof({cname}));')
        for f in fields:
            lines.append(f'printf("{cname}.{f} %zu\\n", offsetof({cname}, {f}));')
    lines += ["return 0;}"]
    src = tmp_path / "abi.c"
    src.write_text("\n".join(lines))
    exe = tmp_path / "abi"
    subprocess.check_call(["gcc", "-I", os.path.join(ROOT, "include"), str(src), "-o", str(exe)])
    got = dict(l.split() for l in subprocess.check_output([str(exe)]).decode().splitlines())
    for cname, (cls, fields) in checks.items():
        assert int(got[cname]) == ctypes.sizeof(cls), cname
        for f in fields:
            assert int(got[f"{cname}.{f}"]) == getattr(cls, f).offset, f"{cname}.{f}"


def test_bad_arguments_are_refused_with_a_message():
    lib = hip.load()
    assert lib.sopro_gemm_f32(None, None) == -2
    assert b"args is NULL" in lib.sopro_last_error()
    g = hip.GemmArgs()
    g.M, g.N, g.K, g.rows_per_seg = 4, 4, 6, 4  # K not a multiple of 4
    assert lib.sopro_gemm_f32(ctypes.byref(g), None) == -2
    assert b"multiple of 4" in lib.sopro_last_error()
    a = hip.SkinnyArgs()
    a.B, a.N, a.K = 1, 16, 48
    assert lib.sopro_skinny_f32(ctypes.byref(a), None) == -2
    at = hip.AttnArgs()
    assert lib.sopro_attention_f32(ctypes.byref(at), None) == -2
    assert lib.sopro_capture_begin(None) == -2


def test_packed_layout_sizes():
    """Host-side size functions of the two packed weight layouts (no device needed)."""
    lib = hip.load()
    # split-bf16 GEMM weights: [n/32][k/16][piece][64 lanes][16 B], K padded to 32
    assert lib.sopro_packed_w_bytes(64, 32, 2) == 2 * 2 * 2 * 64 * 16
    assert lib.sopro_packed_w_bytes(65, 33, 3) == 3 * 4 * 3 * 64 * 16
    assert lib.sopro_packed_w_bytes(64, 32, 4) == 0 and lib.sopro_packed_w_bytes(0, 32, 2) == 0
    # AR-step weights: [column tile][K/32][512 floats]; GLU tiles pair 8 value rows with 8 gate rows
    assert lib.sopro_skinny_packed_floats(1536, 384, 0) == 96 * 12 * 512
    assert lib.sopro_skinny_packed_floats(2049, 384, 0) == 129 * 12 * 512
    assert lib.sopro_skinny_packed_floats(768, 384, 1) == 48 * 12 * 512
    assert lib.sopro_skinny_packed_floats(768, 100, 0) == 0 and lib.sopro_skinny_packed_floats(767, 384, 1) == 0
    assert lib.sopro_pack_skinny_w(None, 0, 16, 32, 0, None, None) == -2


def test_engine_refuses_to_run_without_the_hip_device():
    import torch

    if torch.cuda.is_available():
        pytest.skip("GPU present")
    from sopro_amd.config import SoproTTSConfig
    from sopro_amd.model import SoproTTSModel

    with pytest.raises(hip.SoproHipError):
        SoproTTSModel(SoproTTSConfig(), {}, "cuda:0")
    with pytest.raises(hip.SoproHipError):
        hip.ptr(torch.zeros(4))
