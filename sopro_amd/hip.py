"""ctypes binding of ``libsopro_hip.so`` (C ABI declared in ``include/sopro_hip.h``).

The library is the product: there is no torch / CPU fallback behind these wrappers.  If the
shared object is missing, or a wrapper is handed a tensor that is not a contiguous CUDA(HIP)
tensor, the call raises.  Every wrapper enqueues on torch's *current* stream so that torch's
caching allocator and the kernels agree on ordering (wrap a region in
``torch.cuda.stream(s)`` to move it, e.g. for hipGraph capture).
"""
from __future__ import annotations

import ctypes as C
import gc
import os
import threading
from typing import Optional

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libsopro_hip.so")
ABI_VERSION = 41
# Host-side A/B switches whose decision is on record (profiles/rNN_experiments.md) are read from the environment only in DEVELOPER
# mode (SOPRO_DEV=1) - round 6 housekeeping, the Python twin of the library's `make DEV=1`: a product process ignores them.
DEV_MODE = os.environ.get("SOPRO_DEV", "0") == "1"


def dev_env(name: str, default: str) -> str:
    return os.environ.get(name, default) if DEV_MODE else default


DEFAULT_GROUP_M = 8  # tile-walk group of the split-bf16 contractions (sopro_gemm_set_group_m); measured in tools/pipeline_sweep.sh
PRO_NONE, PRO_ELU, PRO_ADDVEC = 0, 1, 2
EPI_ROPE = 6  # rotate-half RoPE of the leading columns in the contraction's epilogue (sopro_gemm_split_ext.rope_*)
EPI_NONE, EPI_GELU, EPI_GLU, EPI_RES, EPI_TANH, EPI_GLU_DW = 0, 1, 2, 3, 4, 5
NORM_RMS, NORM_LN = 0, 1

_p = C.c_void_p
_i32, _i64, _f32 = C.c_int32, C.c_int64, C.c_float


class GemmArgs(C.Structure):
    _fields_ = [("A", _p), ("lda", _i64), ("a_seg_stride", _i64), ("W", _p), ("ldw", _i64), ("bias", _p),
                ("C", _p), ("ldc", _i64), ("c_seg_stride", _i64), ("R", _p), ("ldr", _i64), ("r_seg_stride", _i64),
                ("scale", _p), ("pro_vec", _p), ("dbg", _p), ("M", _i32), ("N", _i32), ("K", _i32), ("rows_per_seg", _i32),
                ("prologue", _i32), ("epilogue", _i32)]


class SplitExt(C.Structure):
    _fields_ = [("a_format", _i32), ("c_mode", _i32), ("C2", _p), ("ldc2", _i64), ("c2_seg_stride", _i64),
                ("rms_norm", _i32), ("rms_eps", _f32), ("ksplit", _i32), ("n_tickets", _i32), ("ws", _p), ("ws_bytes", _i64),
                ("tickets", _p), ("group_m", _i32), ("acc_scale", _f32), ("range_events", _p), ("rope_cos", _p), ("rope_sin", _p),
                ("rope_cols", _i32), ("rope_dh", _i32), ("rope_pos0", _i32), ("rope_rows_per_seg", _i32), ("ln_stats", _p), ("ln_stats_out", _p)]


class SkinnyArgs(C.Structure):
    _fields_ = [("X", _p), ("ldx", _i64), ("W", _p), ("ldw", _i64), ("bias", _p), ("Y", _p), ("ldy", _i64),
                ("R", _p), ("ldr", _i64), ("scale", _p), ("ring", _p), ("dw_w", _p), ("dw_b", _p), ("step", _p),
                ("Xp", _p), ("xp_stride", _i64), ("y_part_stride", _i64), ("dbg", _p), ("eps", _f32),
                ("B", _i32), ("N", _i32), ("K", _i32), ("epilogue", _i32),
                ("ring_len", _i32), ("ring_bcap", _i32), ("dil", _i32), ("ksize", _i32),
                ("np", _i32), ("ksplit", _i32), ("rms_norm", _i32), ("w_layout", _i32),
                ("aux_W", _p), ("aux_bias", _p), ("aux_R", _p), ("aux_Y", _p), ("aux_ldy", _i64), ("aux_ldr", _i64), ("aux_y_part_stride", _i64),
                ("aux_tiles", _i32), ("aux_flags", _i32), ("ring_format", _i32), ("mt", _i32), ("nt", _i32)]


class AttnArgs(C.Structure):
    _fields_ = [("Q", _p), ("ldq", _i64), ("q_bstride", _i64), ("K", _p), ("ldk", _i64), ("k_bstride", _i64),
                ("V", _p), ("ldv", _i64), ("v_bstride", _i64), ("O", _p), ("ldo", _i64), ("o_bstride", _i64),
                ("klens", _p), ("B", _i32), ("H", _i32), ("dh", _i32), ("Tq", _i32), ("Tk", _i32),
                ("causal", _i32), ("q_pos0", _i32), ("k_pos0", _i32), ("window", _i32), ("scale", _f32), ("kv_index", _p)]


class XattnArgs(C.Structure):
    _fields_ = [("X", _p), ("ldx", _i64), ("Xp", _p), ("xp_stride", _i64), ("norm_w", _p), ("Kp", _p), ("Vp", _p), ("klens", _p),
                ("Y", _p), ("y_part_stride", _i64), ("eps", _f32), ("gate", _f32), ("scale", _f32),
                ("np", _i32), ("B", _i32), ("H", _i32), ("D", _i32), ("S_cap", _i32),
                ("Qp", _p), ("qp_stride", _i64), ("nqp", _i32), ("k_unfolded", _i32), ("kv_format", _i32)]


class ArState(C.Structure):
    _fields_ = [("x_cur", _p), ("cond", _p), ("emb", _p), ("hist", _p), ("step", _p), ("row_step", _p),
                ("first_eos", _p), ("stop_t", _p), ("n_stopped", _p), ("recent", _p), ("params", _p), ("seed", C.c_uint64),
                ("B", _i32), ("D", _i32), ("Tar", _i32), ("max_steps", _i32), ("V", _i32), ("bos_row", _i32),
                ("start", _p), ("row_max", _p), ("row_params", _p), ("nonce", _p), ("row_id", _p), ("key", _p), ("dbg", _p)]


AR_MAX_LAYERS = 16


class ArBlock(C.Structure):
    """sopro_ar_block"""
    _fields_ = [("glu_w", _p), ("glu_b", _p), ("dw_w", _p), ("dw_b", _p), ("ff1_w", _p), ("ff1_b", _p), ("ff2_w", _p), ("ff2_b", _p),
                ("ring", _p), ("kp", _p), ("vp", _p), ("dil", _i32), ("xattn", _i32), ("gate", _f32), ("pad_", _i32),
                ("qa_w", _p), ("qu_w", _p), ("q_b", _p)]


class ArFrame(C.Structure):
    """sopro_ar_frame: the buffers of one autoregressive frame (the launch sequence itself is sopro_ar_issue_frame)."""
    _fields_ = [("blk", ArBlock * AR_MAX_LAYERS), ("head_w", _p), ("head_b", _p), ("x0", _p), ("xa", _p), ("xb", _p), ("part", _p),
                ("u", _p), ("xp", _p), ("qa", _p), ("qpart", _p), ("logits", _p), ("klens", _p),
                ("n_layers", _i32), ("B", _i32), ("D", _i32), ("S_cap", _i32), ("V1", _i32), ("H", _i32), ("ksize", _i32), ("w_layout", _i32),
                ("tile_glu", _i32), ("tile_ff1", _i32), ("tile_ff2", _i32), ("tile_head", _i32), ("eps", _f32), ("k_unfold", _i32), ("store_format", _i32),
                ("st", ArState)]


class EngineCfg(C.Structure):
    """sopro_engine_cfg (stage-level entry points)."""
    _fields_ = [("d_model", _i32), ("codebook_size", _i32), ("num_codebooks", _i32), ("nar_head_dim", _i32), ("bos_row", _i32),
                ("n_layers_ar", _i32), ("ar_kernel", _i32), ("ar_dilations", _i32 * 16), ("ar_xattn", _i32 * 16), ("ar_gate", _f32 * 16),
                ("n_layers_nar", _i32), ("nar_kernel", _i32), ("nar_dilations", _i32 * 16),
                ("n_stages", _i32), ("stage_first_cb", _i32 * 8), ("stage_n_cb", _i32 * 8), ("nar_mix", (_f32 * 2) * 8),
                ("nar_prev_cb_weights", _f32 * 64),
                ("mimi_hidden", _i32), ("mimi_codebook_dim", _i32), ("mimi_heads", _i32), ("mimi_head_dim", _i32), ("mimi_layers", _i32),
                ("mimi_window", _i32), ("mimi_inter", _i32), ("mimi_n_ratios", _i32), ("mimi_ratios", _i32 * 8), ("mimi_num_filters", _i32),
                ("mimi_kernel", _i32), ("mimi_res_kernel", _i32), ("mimi_last_kernel", _i32), ("mimi_compress", _i32),
                ("mimi_n_semantic", _i32), ("mimi_rope_positions", _i32), ("mimi_norm_eps", _f32), ("mimi_final_bias", _f32), ("precision", _i32),
                ("n_layers_text", _i32), ("ref_enc_layers", _i32), ("ref_xattn_layers", _i32), ("ref_xattn_heads", _i32), ("sv_student_dim", _i32),
                ("enc_kernel", _i32)]


class NarIO(C.Structure):
    """sopro_nar_io: the refinement's operands where the stages in front of it left them (round 5)."""
    _fields_ = [("cond", _p), ("cond_bstride", _i64), ("cb0", _p), ("cb0_bstride", _i64), ("lens", _p), ("tokens", _p), ("range_out", _p),
                ("safe", _i32)]


class MimiStreamState(C.Structure):
    """sopro_mimi_stream_state"""
    _fields_ = [("kv", _p), ("cap_rows", _i32), ("kv_len", _i32), ("pos", _i32), ("evict", _i32), ("half", _i32)]


# every symbol declared in include/sopro_hip.h: name -> (restype, argtypes)
SYMBOLS = {
    "sopro_last_error": (C.c_char_p, []),
    "sopro_abi_version": (C.c_int, []),
    "sopro_build_flags": (C.c_int, []),
    "sopro_graph_launch_n": (C.c_int, [_p, _p, _i32]),
    "sopro_host_alloc": (C.c_int, [_i64, C.POINTER(_p)]),
    "sopro_host_free": (C.c_int, [_p]),
    "sopro_copy_to_host_async": (C.c_int, [_p, _p, _i64, _p]),
    "sopro_set_lds_floor": (C.c_int, [C.c_int]),
    "sopro_set_host_wait": (C.c_int, [C.c_int]),
    "sopro_device_info": (C.c_int, [C.c_int, C.POINTER(C.c_int)]),
    "sopro_capture_begin": (C.c_int, [_p]),
    "sopro_capture_end": (C.c_int, [_p, C.POINTER(_p)]),
    "sopro_graph_launch": (C.c_int, [_p, _p]),
    "sopro_graph_destroy": (C.c_int, [_p]),
    "sopro_stream_create_cu_range": (C.c_int, [C.c_int, C.c_int, C.POINTER(_p)]),
    "sopro_stream_create_cu_mask": (C.c_int, [_p, _i32, _p]),
    "sopro_stream_destroy": (C.c_int, [_p]),
    "sopro_gemm_f32": (C.c_int, [C.POINTER(GemmArgs), _p]),
    "sopro_gemm_bf16x3": (C.c_int, [_p, _p, _p, _p]),
    "sopro_gemm_bf16x6": (C.c_int, [_p, _p, _p, _p]),
    "sopro_gemm_bf16x1": (C.c_int, [_p, _p, _p, _p]),
    "sopro_pack_w_bf16": (C.c_int, [_p, _i64, _i32, _i32, _i32, _p, _p]),
    "sopro_fill2d_u32": (C.c_int, [_p, _i64, _i32, _i32, C.c_uint32, _p]),
    "sopro_copy2d_u32": (C.c_int, [_p, _i64, _p, _i64, _i32, _i32, _p]),
    "sopro_pack_w_f16x2": (C.c_int, [_p, _i64, _i32, _i32, _f32, _p, _p]),
    "sopro_f16x3_a_scale": (C.c_float, []),
    "sopro_gemm_f16x3": (C.c_int, [_p, _p, _p, _p]),
    "sopro_packed_w_bytes": (C.c_int64, [_i32, _i32, _i32]),
    "sopro_packed_w_rows_bytes": (C.c_int64, [_i32, _i32]),
    "sopro_pack_w_rows_bf16": (C.c_int, [_p, _i64, _i32, _i32, _p, _p]),
    "sopro_gemm_8p_takes": (C.c_int, [_p, _p]),
    "sopro_gemm_bf16x3_8p": (C.c_int, [_p, _p, _p, _p]),
    "sopro_gemm_bf16_set_tile_override": (C.c_int, [C.c_int]),
    "sopro_gemm_set_group_m": (C.c_int, [C.c_int]),
    "sopro_seanet_tail_set_tiles": (C.c_int, [C.c_int]),
    "sopro_seanet_res_set_tiles": (C.c_int, [C.c_int]),
    "sopro_seanet_res128_bf16": (C.c_int, [_p, _i64, _p, _p, _p, _p, _p, _i64, _i32, _i32, _p]),
    "sopro_seanet_up128_bf16": (C.c_int, [_p, _i64, _p, _p, _p, _i64, _i32, _i32, _p]),
    "sopro_seanet_tail_bf16": (C.c_int, [_p, _i64, _p, _p, _p, _p, _p, _f32, _p, _i64, _i32, _i32, _p]),
    "sopro_seanet_up_set_tiles": (C.c_int, [C.c_int]),
    "sopro_seanet_uptail_set_tiles": (C.c_int, [C.c_int]),
    "sopro_seanet_uptail_f32": (C.c_int, [_p, _i64, _p, _p, _p, _p, _p, _p, _p, _f32, _p, _i64, _i32, _i32, _i32, _p]),
    "sopro_seanet_uptail_bf16": (C.c_int, [_p, _i64, _p, _p, _p, _p, _p, _p, _p, _f32, _p, _i64, _i32, _i32, _p]),
    "sopro_seanet_up128_f32": (C.c_int, [_p, _i64, _p, _p, _p, _i64, _i32, _i32, _i32, _p]),
    "sopro_seanet_res128_f32": (C.c_int, [_p, _i64, _p, _p, _p, _p, _p, _i64, _i32, _i32, _p]),
    "sopro_gemm_set_tile_override": (C.c_int, [C.c_int]),
    "sopro_pack_skinny_w": (C.c_int, [_p, _i64, _i32, _i32, _i32, _p, _p]),
    "sopro_pack_skinny_w_bf16": (C.c_int, [_p, _i64, _i32, _i32, _i32, _p, _p]),
    "sopro_skinny_packed_floats": (_i64, [_i32, _i32, _i32]),
    "sopro_skinny_f32": (C.c_int, [C.POINTER(SkinnyArgs), _p]),
    "sopro_row_stats_f32": (C.c_int, [_p, _i64, _i64, _i32, _i32, _i32, _p, _p]),
    "sopro_norm_f32": (C.c_int, [_p, _i64, _i64, _p, _i64, _p, _p, _p, _p, _i32, _i32, _i32, _f32, _i32, _p]),
    "sopro_rms_match_f32": (C.c_int, [_p, _p, _p, _i32, _i32, _p]),
    "sopro_tanh_affine_f32": (C.c_int, [_p, _p, _f32, _f32, _i64, _p]),
    "sopro_add_pos_f32": (C.c_int, [_p, _p, _p, _i32, _i32, _i32, _i32, _p]),
    "sopro_masked_mean_f32": (C.c_int, [_p, _p, _p, _i32, _i32, _i32, _p]),
    "sopro_stats_pool_f32": (C.c_int, [_p, _p, _p, _p, _i32, _i32, _i32, _p]),
    "sopro_l2norm_f32": (C.c_int, [_p, _p, _i32, _i32, _f32, _p]),
    "sopro_dwconv_f32": (C.c_int, [_p, _p, _p, _p, _p, _p, _i32, _i32, _i32, _i32, _i32, _i32, _i32, _p]),
    "sopro_codebook_sum_f32": (C.c_int, [_p, _i32, _p, _p, _p, _i32, _p, _i64, _p, _f32, _f32, _p, _i64, _i64, _i32, _i32, _i32, _p]),
    "sopro_text_embed_f32": (C.c_int, [_p, _p, _p, _i64, _p, _p, _i32, _i32, _i32, _p]),
    "sopro_argmax_rows_f32": (C.c_int, [_p, _i64, _p, _i64, _i32, _i32, _i32, _p]),
    "sopro_argmax_partials_i32": (C.c_int, [_p, _i64, _p, _i64, _i32, _i32, _i32, _i32, _p]),
    "sopro_fir1_f32": (C.c_int, [_p, _i64, _i32, _p, _p, _p, _i64, _i64, _i32, _i32, _i32, _i32, _i32, _i32, _p]),
    "sopro_rvq_assign_f32": (C.c_int, [_p, _i64, _i32, _p, _p, _i64, _i32, _p, _i64, _i32, _p]),
    "sopro_attention_f32": (C.c_int, [C.POINTER(AttnArgs), _p]),
    "sopro_attention_split_bf16": (C.c_int, [C.POINTER(AttnArgs), C.c_int32, _p]),
    "sopro_attn_decode_f32": (C.c_int, [C.POINTER(AttnArgs), _p]),
    "sopro_xattn_step_f32": (C.c_int, [C.POINTER(XattnArgs), _p]),
    "sopro_rope_f32": (C.c_int, [_p, _i64, _p, _p, _i32, _i32, _i32, _i32, _i32, _p]),
    "sopro_upsample2_f32": (C.c_int, [_p, _p, _p, _i64, _i32, _i32, _i32, _p]),
    "sopro_final_conv_f32": (C.c_int, [_p, _i64, _p, _f32, _p, _i64, _i32, _i32, _p]),
    "sopro_seanet_tail_f32": (C.c_int, [_p, _i64, _p, _p, _p, _p, _p, _f32, _p, _i64, _i32, _i32, _p]),
    "sopro_seanet_res128_p_f32": (C.c_int, [_p, _i64, _p, _p, _p, _p, _p, _i64, _i32, _i32, _i32, _p]),
    "sopro_seanet_tail_p_f32": (C.c_int, [_p, _i64, _p, _p, _p, _p, _p, _f32, _p, _i64, _i32, _i32, _i32, _p]),
    "sopro_engine_create": (C.c_int, [C.POINTER(EngineCfg), C.POINTER(_p)]),
    "sopro_engine_set_tensor": (C.c_int, [_p, C.c_char_p, _p, C.POINTER(_i64), _i32]),
    "sopro_engine_finalize": (C.c_int, [_p, _p]),
    "sopro_engine_destroy": (C.c_int, [_p]),
    "sopro_ar_workspace_bytes": (_i64, [_p, _i32, _i32, _i32]),
    "sopro_ar_begin": (C.c_int, [_p, _p, _i32, _p, _p, _p, _i32, _i32, C.POINTER(_f32), C.c_uint64, C.c_uint32, _p]),
    "sopro_ar_run_graph": (C.c_int, [_p, _i32, _p]),
    "sopro_ar_tokens": (C.c_int, [_p, _p, _p, _p, _p]),
    "sopro_nar_workspace_bytes": (_i64, [_p, _i32, _i32]),
    "sopro_nar_refine": (C.c_int, [_p, _p, _p, _i64, _p, _p, _i32, _i32, _p, _p]),
    "sopro_nar_refine_io": (C.c_int, [_p, _p, C.POINTER(NarIO), _i32, _i32, _p]),
    "sopro_nar_seed_i32": (C.c_int, [_p, _i32, _p, _i64, _i32, _i32, _i32, _p]),
    "sopro_mimi_workspace_bytes": (_i64, [_p, _i32, _i32]),
    "sopro_mimi_decode": (C.c_int, [_p, _p, _p, _i32, _i32, _p, _p]),
    "sopro_mimi_chunk_rows": (_i32, [_i32, _i32]),
    "sopro_mimi_decode_parts": (C.c_int, [_p, _p, _p, _i32, _i32, _p, _i32, _p]),
    "sopro_mimi_stream_kv_bytes": (_i64, [_p, _i32]),
    "sopro_mimi_stream_init": (C.c_int, [_p, C.POINTER(MimiStreamState), _p, _i32]),
    "sopro_mimi_stream_trim": (C.c_int, [C.POINTER(MimiStreamState), _i32]),
    "sopro_mimi_decode_stream": (C.c_int, [_p, _p, C.POINTER(MimiStreamState), _p, _i32, _p, _p]),
    "sopro_ar_init": (C.c_int, [C.POINTER(ArState), _p]),
    "sopro_ar_sample": (C.c_int, [C.POINTER(ArState), _p, _i64, _p]),
    "sopro_ar_admit": (C.c_int, [C.POINTER(ArState), _i32, _p]),
    "sopro_ar_issue_frame": (C.c_int, [C.POINTER(ArFrame), _p]),
    "sopro_ar_fold_text": (C.c_int, [_p, _p, _p, _p, _p, _p, _p, _p, _p, _i32, _i32, _i32, _i32, _i32, _f32, _p]),
    "sopro_ar_fold_text_uk": (C.c_int, [_p, _p, _p, _p, _p, _p, _p, _p, _i32, _i32, _i32, _i32, _i32, _f32, _p]),
    "sopro_engine_upload_tensor": (C.c_int, [_p, C.c_char_p, _p, C.POINTER(_i64), _i32, _p]),
    "sopro_checkpoint_open": (C.c_int, [C.c_char_p, C.c_char_p, C.POINTER(_p)]),
    "sopro_checkpoint_close": (C.c_int, [_p]),
    "sopro_checkpoint_count": (_i32, [_p]),
    "sopro_checkpoint_tensor": (C.c_int, [_p, _i32, C.POINTER(C.c_char_p), C.POINTER(_p), C.POINTER(_i64), C.POINTER(_i32)]),
    "sopro_checkpoint_engine_cfg": (C.c_int, [_p, _i32, C.POINTER(EngineCfg)]),
    "sopro_engine_from_checkpoint": (C.c_int, [_p, _i32, _p, C.POINTER(_p)]),
    "sopro_engine_set_ar_tiles": (C.c_int, [_p, _i32, _i32, _i32, _i32]),
    "sopro_cond_workspace_bytes": (_i64, [_p, _i32, _i32, _i32]),
    "sopro_cond_prepare": (C.c_int, [_p, _p, _p, _p, _i32, _p, _p, _p, _p, _i64, _p, _p, _i32, _i32, _i32, _i32, _p, _p, _p, _p]),
    "sopro_film_coeffs": (C.c_int, [_p, _p, _f32, _i32, _p, _p, _p, _p]),
    "sopro_ref_workspace_bytes": (_i64, [_p, _i32]),
    "sopro_ref_prepare": (C.c_int, [_p, _p, _p, _i32, _p, _p, _p, _p]),
    "sopro_mimi_encode_workspace_bytes": (_i64, [_p, _i32, _i32]),
    "sopro_mimi_encode": (C.c_int, [_p, _p, _p, _i32, _i32, _p, _p]),
    "sopro_cvt_f32_bf16": (C.c_int, [_p, _p, _i64, _p]),
    "sopro_cvt_bf16_f32": (C.c_int, [_p, _p, _i64, _p]),
    "sopro_prof_enable": (C.c_int, [C.c_int]),
    "sopro_prof_collect": (C.c_int, [_p, _i32, _p]),
}

_lib: Optional[C.CDLL] = None


class SoproHipError(RuntimeError):
    pass


def load() -> C.CDLL:
    """Load the shared object (once) and type every entry point.  Raises if it is missing."""
    global _lib
    if _lib is not None:
        return _lib
    path = os.environ.get("SOPRO_HIP_LIB") or LIB_PATH  # developer override (ablation builds of tools/micro); same ABI checks
    if not os.path.exists(path):
        raise SoproHipError(
            f"{path} not found: the HIP kernel library is not built. Run `python -c 'import __graft_entry__ as g; g.build()'` "
            "(or `make -C sopro_amd/csrc`). There is no CPU / torch fallback for the Sopro hot path.")
    lib = C.CDLL(path)
    for name, (res, args) in SYMBOLS.items():
        fn = getattr(lib, name)  # AttributeError if the .so does not export a declared symbol
        fn.restype = res
        fn.argtypes = args
    v = lib.sopro_abi_version()
    if v != ABI_VERSION:
        raise SoproHipError(f"{LIB_PATH} has ABI version {v}, the Python host expects {ABI_VERSION}: rebuild it")
    lib.sopro_gemm_set_group_m(int(dev_env("SOPRO_GEMM_GROUP_M", str(DEFAULT_GROUP_M))))
    if dev_env("SOPRO_GEMM_TILE", ""):  # developer A/B: tile shape of every split contraction (1: 128x128, 2: 256x128, 4: 128x64, 5: 64x64)
        lib.sopro_gemm_bf16_set_tile_override(int(dev_env("SOPRO_GEMM_TILE", "0")))
    _lib = lib
    return lib


class ProfRow(C.Structure):  # sopro_prof_row
    _fields_ = [("family", C.c_char * 40), ("launches", C.c_int64), ("gpu_bound", C.c_int64), ("flops", C.c_double),
                ("flops_bound", C.c_double), ("ms_all", C.c_double), ("ms_bound", C.c_double)]


class Profiler:
    """HIP-event timing of the heavy launches, recorded on the stream each launch is enqueued on
    (bench.py's roofline leg).  Families: the GEMM kernels, the attention kernel, AR graph replays.
    A sample whose stream was IDLE when its first event was recorded is host-bound (the span then contains the host's
    time between the record and the launch, not only the kernel): such samples are kept apart, and the family's time is
    extrapolated from the GPU-bound ones."""

    def __init__(self):
        self.rec = []  # (family, flops, ev0, ev1, gpu_bound)

    def begin(self):
        s = torch.cuda.current_stream()
        e = torch.cuda.Event(enable_timing=True)
        e.record(s)
        return (e, not s.query())  # work still queued in front of the launch: the span is GPU time

    def end(self, family: str, flops: float, e0) -> None:
        e1 = torch.cuda.Event(enable_timing=True)
        e1.record(torch.cuda.current_stream())
        self.rec.append((family, flops, e0[0], e1, e0[1]))

    def summary(self) -> dict:
        torch.cuda.synchronize()
        out: dict = {}
        new = lambda: {"ms": 0.0, "launches": 0, "flops": 0.0, "ms_all": 0.0, "gpu_bound": 0, "ms_bound": 0.0, "flops_bound": 0.0}  # noqa: E731
        # the launches of the stage sequences (NAR refinement, Mimi decoding, text folding) are timed inside the library
        rows = (ProfRow * 32)()
        n = C.c_int32(0)
        _check(load().sopro_prof_collect(rows, 32, C.byref(n)), "sopro_prof_collect")
        for r in rows[: n.value]:
            d = out.setdefault(r.family.decode(), new())
            d["launches"] += r.launches
            d["flops"] += r.flops
            d["ms_all"] += r.ms_all
            d["gpu_bound"] += r.gpu_bound
            d["ms_bound"] += r.ms_bound
            d["flops_bound"] += r.flops_bound
        for fam, fl, e0, e1, bound in self.rec:
            d = out.setdefault(fam, new())
            t = e0.elapsed_time(e1)
            d["ms_all"] += t
            d["launches"] += 1
            d["flops"] += fl
            if bound:
                d["gpu_bound"] += 1
                d["ms_bound"] += t
                d["flops_bound"] += fl
        for d in out.values():
            # time of the family: GPU-bound samples scaled to all launches (by flops where the family has them)
            if d["gpu_bound"] * 10 >= d["launches"] and d["ms_bound"] > 0:
                scale = (d["flops"] / d["flops_bound"]) if d["flops_bound"] > 0 else d["launches"] / d["gpu_bound"]
                d["ms"] = d["ms_bound"] * scale
            else:
                d["ms"] = d["ms_all"]
        return out


_prof: Optional[Profiler] = None
phase_log = None  # bench.py: list receiving (frames, rows, ev0, ev1) per AR phase while set


def set_profiler(p: Optional[Profiler]) -> None:
    global _prof
    _prof = p
    load().sopro_prof_enable(1 if p is not None else 0)


def _check(rc: int, what: str) -> None:
    if rc != 0:
        raise SoproHipError(f"{what} failed ({rc}): {load().sopro_last_error().decode()}")


def _stream() -> int:
    return torch.cuda.current_stream().cuda_stream  # ~1 us; the engines enqueue a few hundred launches per call


def ptr(t: Optional[torch.Tensor], dtype: torch.dtype = torch.float32) -> Optional[int]:
    """Device address of a tensor after checking that it is what the ABI expects."""
    if t is None:
        return None
    if not t.is_cuda:
        raise SoproHipError("tensor is not on a HIP device: the Sopro hot path has no CPU fallback")
    if t.dtype != dtype:
        raise SoproHipError(f"expected {dtype}, got {t.dtype}")
    return t.data_ptr()


# ------------------------------------------------------------------------------------------
# op wrappers (thin: argument marshalling only)
# ------------------------------------------------------------------------------------------
def gemm(A: torch.Tensor, W, Cout: torch.Tensor, *, M: int, N: int, K: int, lda: Optional[int] = None,
         ldc: Optional[int] = None, bias: Optional[torch.Tensor] = None, epilogue: int = EPI_NONE,
         prologue: int = PRO_NONE, R: Optional[torch.Tensor] = None, ldr: Optional[int] = None,
         scale: Optional[torch.Tensor] = None, pro_vec: Optional[torch.Tensor] = None, rows_per_seg: Optional[int] = None,
         a_seg_stride: int = 0, c_seg_stride: int = 0, r_seg_stride: int = 0, a_off: int = 0, c_off: int = 0,
         r_off: int = 0, ldw: Optional[int] = None, dbg: Optional[torch.Tensor] = None, a_split: bool = False, c_mode: int = 0,
         C2: Optional[torch.Tensor] = None, ldc2: Optional[int] = None, c2_seg_stride: int = 0,
         c2_off: int = 0, rms_eps: float = 0.0, range_events: Optional[torch.Tensor] = None, rope: Optional[tuple] = None,
         ln_stats: Optional[torch.Tensor] = None, ln_eps: float = 0.0, ln_stats_out: Optional[torch.Tensor] = None,
         long_k: Optional[bool] = None) -> None:
    """C = epi(pro(A) @ W^T + bias); offsets are in elements from the tensors' first element.  With a ``PackedW`` weight the
    contraction runs on the split-bf16 path, where ``a_split`` says A is in split form and ``c_mode`` 1 / 2 writes
    ELU(C) in split form (to C, or to C2 next to the fp32 C): see sopro_gemm_split_ext in include/sopro_hip.h."""
    n_out = N // 2 if epilogue == EPI_GLU else N
    g = GemmArgs()
    a16 = A.dtype == torch.bfloat16  # bf16 rows (a_format 2 of sopro_gemm_bf16x1): offsets / strides count bf16 elements
    c16 = c_mode in (6, 7, 8)        # bf16 rows out: Cout / C2 / R are bf16 tensors
    cdt = torch.bfloat16 if c16 else torch.float32
    g.A = ptr(A, A.dtype if a16 else torch.float32) + (2 if a16 else 4) * a_off
    g.lda = K if lda is None else lda
    g.a_seg_stride = a_seg_stride
    packed = isinstance(W, PackedW)
    if packed and (W.N != N or W.K != K):
        raise SoproHipError(f"packed weight is [{W.N}, {W.K}], the call wants [{N}, {K}]")
    g.W = None if packed else ptr(W)
    g.ldw = K if ldw is None else ldw
    g.bias = ptr(bias)
    g.C = (ptr(Cout, cdt) + (2 if c16 else 4) * c_off) if Cout is not None else None  # None only with c_mode 5 (arg-max partials to C2)
    g.ldc = n_out if ldc is None else ldc
    g.c_seg_stride = c_seg_stride
    g.R = (ptr(R, cdt) + (2 if c16 else 4) * r_off) if R is not None else None
    g.ldr = (n_out if ldr is None else ldr)
    g.r_seg_stride = r_seg_stride
    g.scale = ptr(scale)
    g.pro_vec = ptr(pro_vec)
    g.dbg = ptr(dbg, torch.int64)
    g.M, g.N, g.K = M, N, K
    g.rows_per_seg = M if rows_per_seg is None else rows_per_seg
    g.prologue, g.epilogue = prologue, epilogue
    e0 = _prof.begin() if _prof is not None else None
    x = None
    if packed:
        x = SplitExt()
        if rope is not None:  # (cos table, sin table, columns, head dim, first position, rows per utterance): EPI_ROPE
            x.rope_cos, x.rope_sin = ptr(rope[0]), ptr(rope[1])
            x.rope_cols, x.rope_dh, x.rope_pos0, x.rope_rows_per_seg = int(rope[2]), int(rope[3]), int(rope[4]), int(rope[5])
        if ln_stats is not None:  # fused LayerNorm of the A rows from the producer's statistics (W / bias carry the norm's weight / bias)
            x.ln_stats, x.rms_eps = ptr(ln_stats), float(ln_eps)
        x.ln_stats_out = ptr(ln_stats_out)  # EPI_RES: statistics of the updated stream (sopro_gemm_split_ext.ln_stats_out)
        if rms_eps > 0.0:  # fused RMSNorm of the A rows (W must carry the norm's weight vector)
            if W.pieces == 2 and not W.f16:
                raise SoproHipError("fused RMSNorm is a six-pass (pieces = 3), f16 three-pass or one-pass (pieces = 1) feature")
            x.rms_norm, x.rms_eps = 1, float(rms_eps)
        ks = _auto_ksplit(M, N, K, 3 if W.f16 else W.pieces, epilogue) if (dbg is None and _splitk_enabled and rms_eps <= 0.0) else 1
        if ks > 1:
            ws, tk = _splitk_buffers()
            x.ksplit, x.n_tickets, x.ws, x.ws_bytes, x.tickets = ks, int(tk.numel()), ptr(ws), int(ws.numel()) * 4, ptr(tk, torch.int32)
    if (a16 or c16) and not (packed and W.pieces == 1 and not W.f16):
        raise SoproHipError("bf16 rows (a_format 2, c_mode 6 / 7 / 8) belong to the one-pass (pieces = 1) path")
    if packed and W.f16:  # two fp16 pieces, three passes: the forms of the six-pass path
        if a_split or c_mode not in (0, 5):
            raise SoproHipError("split-form operands belong to the bf16 three-pass (pieces = 2) path")
        if c_mode == 5:
            x.c_mode, x.C2, x.ldc2 = 5, ptr(C2) + 4 * c2_off, (-(-N // 64) if ldc2 is None else ldc2)
        x.acc_scale = W.acc_scale
        x.range_events = ptr(range_events, torch.int32)  # optional device word (sopro_gemm_split_ext.range_events)
        _check(load().sopro_gemm_f16x3(C.byref(g), ptr(W.data, torch.int32), C.byref(x), _stream()), "sopro_gemm_f16x3")
    elif packed and W.pieces == 1:
        if a_split or c_mode in (1, 2):
            raise SoproHipError("split-form operands belong to the three-pass (pieces = 2) path")
        x.c_mode = c_mode
        x.a_format = 2 if a16 else 0
        x.C2 = (ptr(C2, cdt) + (2 if c16 else 4) * c2_off) if C2 is not None else None
        x.ldc2, x.c2_seg_stride = (n_out if ldc2 is None else ldc2), c2_seg_stride
        _check(load().sopro_gemm_bf16x1(C.byref(g), ptr(W.data, torch.int32), C.byref(x), _stream()), "sopro_gemm_bf16x1")
    elif packed and W.pieces == 3:
        if a_split or c_mode not in (0, 5):
            raise SoproHipError("split-form operands belong to the three-pass (pieces = 2) path")
        if c_mode == 5:
            x.c_mode, x.C2, x.ldc2 = 5, ptr(C2) + 4 * c2_off, (-(-N // 64) if ldc2 is None else ldc2)
        _check(load().sopro_gemm_bf16x6(C.byref(g), ptr(W.data, torch.int32), C.byref(x), _stream()), "sopro_gemm_bf16x6")
    elif packed:
        if a16 or c16:
            raise SoproHipError("bf16 rows (a_format 2, c_mode 6 / 7 / 8) belong to the one-pass (pieces = 1) path")
        x.a_format, x.c_mode = int(bool(a_split)), c_mode
        x.C2 = (ptr(C2) + 4 * c2_off) if C2 is not None else None
        x.ldc2, x.c2_seg_stride = (n_out if ldc2 is None else ldc2), c2_seg_stride
        use_8p = long_k if long_k is not None else (W.rows is not None and bool(load().sopro_gemm_8p_takes(C.byref(g), C.byref(x))))
        if use_8p:  # the long-K form: 256 x 256 tiles, both operands split in memory (csrc/gemm_8p.hip); same results bit for bit
            if W.rows is None:
                raise SoproHipError("the long-K form needs the weight's split-form rows (pack_w_bf16x3(W, rows=True))")
            x.ksplit = 0
            _check(load().sopro_gemm_bf16x3_8p(C.byref(g), W.rows.data_ptr(), C.byref(x), _stream()), "sopro_gemm_bf16x3_8p")
        else:
            _check(load().sopro_gemm_bf16x3(C.byref(g), ptr(W.data, torch.int32), C.byref(x), _stream()), "sopro_gemm_bf16x3")
    elif a_split or c_mode:
        raise SoproHipError("split-plane operands need a PackedW weight (the fp32 kernel reads and writes fp32 rows)")
    else:
        _check(load().sopro_gemm_f32(C.byref(g), _stream()), "sopro_gemm_f32")
    if e0 is not None:
        fam = "gemm_f32_kernel" if not packed else ("gemm_f16x3_kernel" if W.f16 else {1: "gemm_bf16x1_kernel", 2: "gemm_bf16x3_kernel", 3: "gemm_bf16x6_kernel"}[W.pieces])
        _prof.end(fam, 2.0 * M * N * K, e0)


_splitk_enabled = os.environ.get("SOPRO_NO_SPLITK", "0") != "1"
_splitk_pool: dict = {}  # stream handle -> (workspace, tickets): split-K scratch belongs to one stream at a time
_SPLITK_WS_BYTES = 32 << 20
_SPLITK_TICKETS = 1024


def _auto_ksplit(M: int, N: int, K: int, pieces: int, epilogue: int) -> int:
    """K slices for a problem with too few output tiles to occupy the chip (streaming chunks, batch 1): its K loop
    would otherwise run at one memory latency per 32-wide step on a handful of workgroups."""
    if pieces == 3:
        bm, bn = (64, 128) if epilogue == EPI_GLU else (64, 64)
    elif pieces == 1:  # sopro_gemm_bf16x1's tile rule
        small = N <= 64 or M <= 64 or (-(-M // 128) * -(-N // 128)) < 256
        bm, bn = ((64, 128) if epilogue == EPI_GLU else (64, 64)) if small else (128, 128)
    else:
        bm, bn = (64, 64) if (N <= 64 or M <= 64) else (128, 128)
    tiles = -(-M // bm) * -(-N // bn)
    kt = -(-K // 32)
    # the split costs ~10 us (device-scope release / acquire around the ticket): it pays from ~32 K-steps (K >= 1024) up
    if tiles >= 96 or kt < 32 or tiles > _SPLITK_TICKETS:
        return 1
    # a slice costs ~0.9 us per K-step, the reducing workgroup ~0.36 us per slice of a 64x64 tile: ks ~ sqrt(2.5 * kt)
    ks = max(1, min(16, int(round((2.5 * kt) ** 0.5)), 512 // tiles))
    while ks > 1 and ks * tiles * bm * bn * 4 > _SPLITK_WS_BYTES:
        ks -= 1
    return ks


def _splitk_buffers():
    key = _stream()
    bufs = _splitk_pool.get(key)
    if bufs is None:
        dev = torch.cuda.current_device()
        bufs = (torch.empty(_SPLITK_WS_BYTES // 4, dtype=torch.float32, device=f"cuda:{dev}"),
                torch.zeros(_SPLITK_TICKETS, dtype=torch.int32, device=f"cuda:{dev}"))
        _splitk_pool[key] = bufs
    return bufs


class PackedW:
    """A weight matrix [N, K] as 1, 2 or 3 bf16 pieces in MFMA fragment order (sopro_pack_w_bf16)."""

    __slots__ = ("data", "N", "K", "pieces", "f16", "acc_scale", "rows")

    def __init__(self, data: torch.Tensor, N: int, K: int, pieces: int, f16: bool = False, acc_scale: float = 1.0):
        self.data, self.N, self.K, self.pieces, self.f16, self.acc_scale = data, N, K, pieces, f16, acc_scale
        self.rows = None  # pieces == 2: the same matrix as split-form ROWS for the long-K form (pack_w_rows_bf16 / sopro_gemm_bf16x3_8p)


def pack_w_bf16(W: torch.Tensor, pieces: int) -> PackedW:
    """[N, K] fp32 device matrix -> the operand of ``gemm`` on the split-bf16 matrix-core paths
    (pieces = 2: three passes, 16 mantissa bits; pieces = 3: six passes, 24 mantissa bits)."""
    if W.dim() != 2 or not W.is_contiguous():
        raise SoproHipError("pack_w_bf16 wants a contiguous [N, K] matrix")
    N, K = int(W.shape[0]), int(W.shape[1])
    lib = load()
    data = torch.empty(int(lib.sopro_packed_w_bytes(N, K, pieces)) // 4, dtype=torch.int32, device=W.device)
    _check(lib.sopro_pack_w_bf16(ptr(W), K, N, K, pieces, ptr(data, torch.int32), _stream()), "sopro_pack_w_bf16")
    return PackedW(data, N, K, pieces)


def pack_w_bf16x3(W: torch.Tensor, rows: bool = False) -> PackedW:
    """``rows``: also keep the matrix as split-form rows [N][K / 32][32 hi | 32 lo] (sopro_pack_w_rows_bf16) - the weight operand of the
    long-K form (csrc/gemm_8p.hip), which ``gemm`` picks when the library says the call is one it takes (or ``long_k=True``)."""
    pw = pack_w_bf16(W, 2)
    if rows:
        N, K = pw.N, pw.K
        lib = load()
        nb = int(lib.sopro_packed_w_rows_bytes(N, K))
        if nb <= 0:
            raise SoproHipError("split-form weight rows need K % 32 == 0")
        buf = torch.empty(nb // 4 + 32, dtype=torch.int32, device=W.device)
        off = (-buf.data_ptr() % 128) // 4  # 128-byte aligned rows
        pw.rows = buf[off: off + nb // 4]
        _check(lib.sopro_pack_w_rows_bf16(ptr(W), K, N, K, pw.rows.data_ptr(), _stream()), "sopro_pack_w_rows_bf16")
    return pw


def pack_w_f16x3(W: torch.Tensor) -> PackedW:
    """Two fp16 pieces (22 mantissa bits), three MFMA passes: the token paths' accuracy class at half the passes of bf16x6.
    The matrix is scaled by a power of two into fp16's range (max |w| lands in [2^13, 2^14)); gemm undoes it exactly."""
    if W.dim() != 2 or not W.is_contiguous():
        raise SoproHipError("pack_w_f16x3 wants a contiguous [N, K] matrix")
    import math

    N, K = int(W.shape[0]), int(W.shape[1])
    lib = load()
    amax = float(W.abs().max())
    wscale = 2.0 ** (13 - math.frexp(amax)[1] + 1) if amax > 0.0 else 1.0  # amax * wscale in [2^13, 2^14)
    data = torch.empty(int(lib.sopro_packed_w_bytes(N, K, 2)) // 4, dtype=torch.int32, device=W.device)
    _check(lib.sopro_pack_w_f16x2(ptr(W), K, N, K, wscale, ptr(data, torch.int32), _stream()), "sopro_pack_w_f16x2")
    return PackedW(data, N, K, 2, f16=True, acc_scale=1.0 / (float(lib.sopro_f16x3_a_scale()) * wscale))


def pack_w_bf16x6(W: torch.Tensor) -> PackedW:
    return pack_w_bf16(W, 3)


def pack_w_bf16x1(W: torch.Tensor) -> PackedW:
    """bf16 mode: the weight rounded to bf16 once (one MFMA pass per product)."""
    return pack_w_bf16(W, 1)


class SkinnyW:
    """AR-step weights in the fragment order of ``sopro_pack_skinny_w`` (every load instruction reads 1 KiB of consecutive memory)."""

    __slots__ = ("data", "N", "K", "glu", "bf16")

    def __init__(self, data: torch.Tensor, N: int, K: int, glu: bool, bf16: bool = False):
        self.data, self.N, self.K, self.glu, self.bf16 = data, N, K, glu, bf16


def pack_skinny_w(W: torch.Tensor, glu: bool = False, bf16: bool = False) -> SkinnyW:
    """``bf16``: the engine's bf16 mode - weights rounded to bf16 (half the bytes), one bf16 MFMA per 32-wide K chunk."""
    N, K = int(W.shape[0]), int(W.shape[1])
    n = int(load().sopro_skinny_packed_floats(N, K, int(glu)))
    if n <= 0:
        raise SoproHipError(f"pack_skinny_w: unsupported shape {N}x{K} (K % 32 == 0; glu: N even)")
    Wc = W.contiguous()
    with torch.cuda.device(W.device):
        if bf16:
            out = torch.empty(n // 2, dtype=torch.int32, device=W.device)
            _check(load().sopro_pack_skinny_w_bf16(ptr(Wc), K, N, K, int(glu), ptr(out, torch.int32), _stream()), "sopro_pack_skinny_w_bf16")
        else:
            out = torch.empty(n, dtype=torch.float32, device=W.device)
            _check(load().sopro_pack_skinny_w(ptr(Wc), K, N, K, int(glu), ptr(out), _stream()), "sopro_pack_skinny_w")
    return SkinnyW(out, N, K, bool(glu), bool(bf16))


def skinny(X: torch.Tensor, W, Y: torch.Tensor, *, B: int, N: int, K: int, ldx: Optional[int] = None,
           ldy: Optional[int] = None, rms_norm: bool = False, eps: float = 1e-6, bias: Optional[torch.Tensor] = None,
           epilogue: int = EPI_NONE, R: Optional[torch.Tensor] = None, ldr: Optional[int] = None,
           scale: Optional[torch.Tensor] = None, ring: Optional[torch.Tensor] = None, dw_w: Optional[torch.Tensor] = None,
           dw_b: Optional[torch.Tensor] = None, step: Optional[torch.Tensor] = None, ring_len: int = 0, ring_bcap: int = 0,
           dil: int = 1, ksize: int = 1, Xp: Optional[torch.Tensor] = None, np_: int = 0, xp_stride: int = 0,
           ksplit: bool = False, y_part_stride: int = 0, dbg: Optional[torch.Tensor] = None, mt: int = 1, nt: int = 1,
           aux_W=None, aux_Y: Optional[torch.Tensor] = None, aux_bias: Optional[torch.Tensor] = None, aux_R: Optional[torch.Tensor] = None,
           aux_flags: int = 0, aux_y_part_stride: int = 0) -> None:
    """AR-step contraction.  With rms_norm the RMSNorm weight must already be folded into W (W * w_norm[None, :]).
    ``mt`` x ``nt``: 16-row groups x column tiles per workgroup (1 or 2 each; results are bit-identical)."""
    a = SkinnyArgs()
    a.X, a.ldx = ptr(X), (K if ldx is None else ldx)
    if isinstance(W, SkinnyW):
        if (W.N, W.K, W.glu) != (N, K, epilogue == EPI_GLU_DW):
            raise SoproHipError(f"packed weight is {W.N}x{W.K} glu={W.glu}, the call says {N}x{K} glu={epilogue == EPI_GLU_DW}")
        a.W, a.ldw, a.w_layout = (ptr(W.data, torch.int32), K, 2) if W.bf16 else (ptr(W.data), K, 1)
    else:
        a.W, a.ldw = ptr(W), K
    a.bias = ptr(bias)
    n_out = N // 2 if epilogue == EPI_GLU_DW else N
    a.Y, a.ldy = ptr(Y), (n_out if ldy is None else ldy)
    a.R, a.ldr, a.scale = ptr(R), (n_out if ldr is None else ldr), ptr(scale)
    ring_bf16 = ring is not None and ring.dtype == torch.bfloat16
    a.ring, a.dw_w, a.dw_b, a.step = ptr(ring, torch.bfloat16 if ring_bf16 else torch.float32), ptr(dw_w), ptr(dw_b), ptr(step, torch.int32)
    a.ring_format = 1 if ring_bf16 else 0
    a.Xp, a.xp_stride, a.y_part_stride, a.dbg = ptr(Xp), xp_stride, y_part_stride, ptr(dbg, torch.int64)
    a.eps = eps
    a.B, a.N, a.K, a.epilogue = B, N, K, epilogue
    a.ring_len, a.ring_bcap, a.dil, a.ksize = ring_len, ring_bcap, dil, ksize
    a.np, a.ksplit, a.rms_norm = np_, int(ksplit), int(rms_norm)
    a.mt, a.nt = int(mt), int(nt)
    if aux_W is not None:  # aux column tiles (sopro_skinny_args.aux_*): a second operand set on the same input rows
        if not isinstance(aux_W, SkinnyW) or aux_W.K != K or aux_W.glu or aux_W.bf16 != (a.w_layout == 2):
            raise SoproHipError("aux_W must be a packed (non-GLU) weight of the same K and element type as W")
        a.aux_W = ptr(aux_W.data, torch.int32) if aux_W.bf16 else ptr(aux_W.data)
        a.aux_tiles, a.aux_flags = (aux_W.N + 15) // 16, int(aux_flags)
        a.aux_Y, a.aux_ldy, a.aux_bias, a.aux_R, a.aux_ldr = ptr(aux_Y), aux_W.N, ptr(aux_bias), ptr(aux_R), aux_W.N
        a.aux_y_part_stride = aux_y_part_stride
    _check(load().sopro_skinny_f32(C.byref(a), _stream()), "sopro_skinny_f32")


def norm(x: torch.Tensor, out: torch.Tensor, w: torch.Tensor, *, rows: int, C_: int, eps: float, kind: int = NORM_RMS,
         b: Optional[torch.Tensor] = None, mul: Optional[torch.Tensor] = None, add: Optional[torch.Tensor] = None,
         rows_per_seg: Optional[int] = None, ldx: Optional[int] = None, ldo: Optional[int] = None, x_off: int = 0,
         o_off: int = 0, x_seg_stride: int = 0) -> None:
    _check(load().sopro_norm_f32(ptr(x) + 4 * x_off, C_ if ldx is None else ldx, x_seg_stride, ptr(out) + 4 * o_off,
                                 C_ if ldo is None else ldo, ptr(w), ptr(b), ptr(mul), ptr(add), rows,
                                 rows if rows_per_seg is None else rows_per_seg, C_, eps, kind, _stream()), "sopro_norm_f32")


def row_stats(x: torch.Tensor, rows: int, C_: int, stats: torch.Tensor, ldx: Optional[int] = None, x_seg_stride: int = 0,
              rows_per_seg: Optional[int] = None, x_off: int = 0) -> None:
    """stats [rows, C / 64, 2] = (mean, squared deviations) per 64-column group: what a contraction with ``ln_stats`` stages by."""
    _check(load().sopro_row_stats_f32(ptr(x) + 4 * x_off, C_ if ldx is None else ldx, x_seg_stride, rows, rows if rows_per_seg is None else rows_per_seg,
                                      C_, ptr(stats), _stream()), "sopro_row_stats_f32")


def rms_match(a: torch.Tensor, x: torch.Tensor, out: torch.Tensor, rows: int, C_: int) -> None:
    _check(load().sopro_rms_match_f32(ptr(a), ptr(x), ptr(out), rows, C_, _stream()), "sopro_rms_match_f32")


def tanh_affine(x: torch.Tensor, out: torch.Tensor, c0: float, c1: float, n: int) -> None:
    _check(load().sopro_tanh_affine_f32(ptr(x), ptr(out), c0, c1, n, _stream()), "sopro_tanh_affine_f32")


def add_pos(rowvec: torch.Tensor, table: torch.Tensor, out: torch.Tensor, B: int, T: int, C_: int, pos0: int = 0) -> None:
    _check(load().sopro_add_pos_f32(ptr(rowvec), ptr(table), ptr(out), B, T, C_, pos0, _stream()), "sopro_add_pos_f32")


def masked_mean(x: torch.Tensor, lens: Optional[torch.Tensor], out: torch.Tensor, B: int, T: int, C_: int) -> None:
    _check(load().sopro_masked_mean_f32(ptr(x), ptr(lens, torch.int32), ptr(out), B, T, C_, _stream()), "sopro_masked_mean_f32")


def stats_pool(h: torch.Tensor, logit: torch.Tensor, lens: Optional[torch.Tensor], out: torch.Tensor, B: int, T: int, C_: int) -> None:
    _check(load().sopro_stats_pool_f32(ptr(h), ptr(logit), ptr(lens, torch.int32), ptr(out), B, T, C_, _stream()), "sopro_stats_pool_f32")


def l2norm(x: torch.Tensor, out: torch.Tensor, rows: int, C_: int, eps: float = 1e-6) -> None:
    _check(load().sopro_l2norm_f32(ptr(x), ptr(out), rows, C_, eps, _stream()), "sopro_l2norm_f32")


def dwconv(x: torch.Tensor, w: torch.Tensor, bias: Optional[torch.Tensor], out: torch.Tensor, *, B: int, T: int, C_: int,
           ksize: int, dil: int, left: int, mode: int = 0, res: Optional[torch.Tensor] = None,
           lens: Optional[torch.Tensor] = None) -> None:
    _check(load().sopro_dwconv_f32(ptr(x), ptr(w), ptr(bias), ptr(res), ptr(out), ptr(lens, torch.int32), B, T, C_, ksize,
                                   dil, left, mode, _stream()), "sopro_dwconv_f32")


def codebook_sum(tok: torch.Tensor, ldt: int, col: torch.Tensor, off: torch.Tensor, wq: torch.Tensor, table: torch.Tensor,
                 out: torch.Tensor, *, rows: int, D: int, base: Optional[torch.Tensor] = None, alpha: float = 0.0,
                 beta: float = 1.0, ldo: Optional[int] = None, rows_per_seg: Optional[int] = None, o_seg_stride: int = 0,
                 o_off: int = 0, tok_off: int = 0) -> None:
    _check(load().sopro_codebook_sum_f32(ptr(tok, torch.int32) + 4 * tok_off, ldt, ptr(col, torch.int32), ptr(off, torch.int32),
                                         ptr(wq), int(col.numel()), ptr(table), int(table.shape[0]), ptr(base), alpha, beta,
                                         ptr(out) + 4 * o_off, D if ldo is None else ldo, o_seg_stride, rows,
                                         rows if rows_per_seg is None else rows_per_seg, D, _stream()), "sopro_codebook_sum_f32")


def text_embed(ids: torch.Tensor, lens: Optional[torch.Tensor], table: torch.Tensor, pe: torch.Tensor, out: torch.Tensor,
               B: int, T: int, C_: int) -> None:
    _check(load().sopro_text_embed_f32(ptr(ids, torch.int32), ptr(lens, torch.int32), ptr(table), int(table.shape[0]), ptr(pe),
                                       ptr(out), B, T, C_, _stream()), "sopro_text_embed_f32")


def argmax_rows(x: torch.Tensor, out: torch.Tensor, *, rows: int, N: int, ldx: Optional[int] = None, ldo: int = 1,
                o_off: int = 0, inner: int = 1) -> None:
    """out[(r // inner) * ldo + r % inner] = argmax(x[r, :N]): ``inner`` consecutive rows share one output row."""
    _check(load().sopro_argmax_rows_f32(ptr(x), N if ldx is None else ldx, ptr(out, torch.int32) + 4 * o_off, ldo, inner, rows, N,
                                        _stream()), "sopro_argmax_rows_f32")


def argmax_partials(part: torch.Tensor, out: torch.Tensor, *, rows: int, heads: int, per_head: int, V: int, ldp: int, ldo: int = 1,
                    o_off: int = 0) -> None:
    """Finish the arg-max that ``gemm(..., c_mode=5, C2=part)`` started: out[r * ldo + h] = token of head h of row r."""
    _check(load().sopro_argmax_partials_i32(ptr(part), ldp, ptr(out, torch.int32) + 4 * o_off, ldo, heads, per_head, V, rows, _stream()),
           "sopro_argmax_partials_i32")


def fir1(x: torch.Tensor, w: torch.Tensor, out: torch.Tensor, *, B: int, n_in: int, n_out: int, C_: int, K: int, stride: int,
         left: int, bias: Optional[torch.Tensor] = None, ldo: Optional[int] = None, x_seg_stride: Optional[int] = None,
         o_seg_stride: Optional[int] = None, o_off: int = 0) -> None:
    ldo = C_ if ldo is None else ldo
    _check(load().sopro_fir1_f32(ptr(x), n_in if x_seg_stride is None else x_seg_stride, n_in, ptr(w), ptr(bias) if bias is not None else None,
                                 ptr(out) + 4 * o_off, ldo, n_out * ldo if o_seg_stride is None else o_seg_stride, B, n_out, C_, K,
                                 stride, left, _stream()), "sopro_fir1_f32")


def rvq_assign(scores: torch.Tensor, table: torch.Tensor, res: torch.Tensor, codes: torch.Tensor, *, rows: int, V: int, D: int,
               ldc: int, t_off: int = 0, c_off: int = 0) -> None:
    _check(load().sopro_rvq_assign_f32(ptr(scores), V, V, ptr(table) + 4 * t_off, ptr(res), D, D, ptr(codes, torch.int32) + 4 * c_off,
                                       ldc, rows, _stream()), "sopro_rvq_assign_f32")


def attention(Q: torch.Tensor, K: torch.Tensor, V: torch.Tensor, O: torch.Tensor, *, B: int, H: int, dh: int, Tq: int,
              Tk: int, ldq: int, ldk: int, ldv: int, ldo: int, q_bstride: int, k_bstride: int, v_bstride: int,
              o_bstride: int, klens: Optional[torch.Tensor] = None, causal: bool = False, q_pos0: int = 0, k_pos0: int = 0,
              window: int = 0, scale: Optional[float] = None, q_off: int = 0, k_off: int = 0, v_off: int = 0,
              o_off: int = 0, decode: bool = False, kv_index: Optional[torch.Tensor] = None, split_passes: int = 0) -> None:
    a = AttnArgs()
    a.kv_index = ptr(kv_index, torch.int32)
    a.Q, a.ldq, a.q_bstride = ptr(Q) + 4 * q_off, ldq, q_bstride
    a.K, a.ldk, a.k_bstride = ptr(K) + 4 * k_off, ldk, k_bstride
    a.V, a.ldv, a.v_bstride = ptr(V) + 4 * v_off, ldv, v_bstride
    a.O, a.ldo, a.o_bstride = ptr(O) + 4 * o_off, ldo, o_bstride
    a.klens = ptr(klens, torch.int32)
    a.B, a.H, a.dh, a.Tq, a.Tk = B, H, dh, Tq, Tk
    a.causal, a.q_pos0, a.k_pos0, a.window = int(causal), q_pos0, k_pos0, window
    a.scale = float(scale if scale is not None else dh ** -0.5)
    if decode:
        _check(load().sopro_attn_decode_f32(C.byref(a), _stream()), "sopro_attn_decode_f32")
        return
    e0 = _prof.begin() if _prof is not None else None
    if split_passes:  # the codec decoder's waveform-path form (two bf16 pieces / three passes, or one)
        _check(load().sopro_attention_split_bf16(C.byref(a), int(split_passes), _stream()), "sopro_attention_split_bf16")
    else:
        _check(load().sopro_attention_f32(C.byref(a), _stream()), "sopro_attention_f32")
    if e0 is not None:
        pairs = float(Tq) * Tk if not causal else float(sum(max(0, min(Tk - 1, q_pos0 + q - k_pos0) - max(0, q_pos0 + q - window + 1 - k_pos0) + 1) for q in range(Tq)))
        _prof.end("attention_kernel", 4.0 * dh * pairs * B * H, e0)


def xattn_step(X: torch.Tensor, Y: torch.Tensor, norm_w: Optional[torch.Tensor], Kp: torch.Tensor, Vp: torch.Tensor, klens: Optional[torch.Tensor], *,
               B: int, H: int, D: int, S_cap: int, gate: float, scale: float, eps: float, Xp: Optional[torch.Tensor] = None, np_: int = 0,
               xp_stride: int = 0, y_part_stride: int = 0, Qp: Optional[torch.Tensor] = None, nqp: int = 0, qp_stride: int = 0) -> None:
    """``Qp`` given: unfolded keys - Kp is K [B, S_cap, D] and Qp the nqp partial sums of the raw query (sopro_xattn_args.k_unfolded)."""
    a = XattnArgs()
    if Qp is not None:
        a.Qp, a.nqp, a.qp_stride, a.k_unfolded = ptr(Qp), int(nqp), int(qp_stride), 1
    a.X, a.ldx, a.Xp, a.xp_stride, a.np = ptr(X), D, ptr(Xp), xp_stride, np_
    kv16 = Kp.dtype == torch.bfloat16
    a.norm_w, a.Kp, a.Vp, a.klens = ptr(norm_w), ptr(Kp, Kp.dtype), ptr(Vp, Kp.dtype), ptr(klens, torch.int32)
    a.kv_format = 1 if kv16 else 0
    a.Y, a.y_part_stride = ptr(Y), y_part_stride
    a.eps, a.gate, a.scale = eps, gate, scale
    a.B, a.H, a.D, a.S_cap = B, H, D, S_cap
    _check(load().sopro_xattn_step_f32(C.byref(a), _stream()), "sopro_xattn_step_f32")


def rope(x: torch.Tensor, cos_t: torch.Tensor, sin_t: torch.Tensor, *, rows: int, rows_per_seg: int, pos0: int, H: int,
         dh: int, ldx: int, x_off: int = 0) -> None:
    _check(load().sopro_rope_f32(ptr(x) + 4 * x_off, ldx, ptr(cos_t), ptr(sin_t), rows, rows_per_seg, pos0, H, dh, _stream()),
           "sopro_rope_f32")


def upsample2(x: torch.Tensor, w: torch.Tensor, y: torch.Tensor, *, B: int, T: int, C_: int, y_seg_stride: int,
              y_off: int = 0) -> None:
    _check(load().sopro_upsample2_f32(ptr(x), ptr(w), ptr(y) + 4 * y_off, y_seg_stride, B, T, C_, _stream()), "sopro_upsample2_f32")


def final_conv(h: torch.Tensor, w: torch.Tensor, bias: float, wav: torch.Tensor, *, B: int, T: int, h_seg_stride: int,
               wav_seg_stride: int) -> None:
    _check(load().sopro_final_conv_f32(ptr(h), h_seg_stride, ptr(w), bias, ptr(wav), wav_seg_stride, B, T, _stream()),
           "sopro_final_conv_f32")


def seanet_tail(h: torch.Tensor, w1: torch.Tensor, b1: torch.Tensor, w2: torch.Tensor, b2: torch.Tensor, wf: torch.Tensor,
                bf: float, wav: torch.Tensor, *, B: int, T: int, h_seg_stride: int, wav_seg_stride: int) -> None:
    _check(load().sopro_seanet_tail_f32(ptr(h), h_seg_stride, ptr(w1), ptr(b1), ptr(w2), ptr(b2), ptr(wf), bf, ptr(wav),
                                        wav_seg_stride, B, T, _stream()), "sopro_seanet_tail_f32")


def seanet_res128(h: torch.Tensor, w1: torch.Tensor, b1: torch.Tensor, w2: torch.Tensor, b2: torch.Tensor, out: torch.Tensor, *, B: int,
                  T: int, h_seg_stride: int, out_seg_stride: int) -> None:
    """Fused 128-channel residual block of the SEANet decoder + the next layer's ELU (sopro_seanet_res128_f32)."""
    _check(load().sopro_seanet_res128_f32(ptr(h), h_seg_stride, ptr(w1), ptr(b1), ptr(w2), ptr(b2), ptr(out), out_seg_stride, B, T,
                                          _stream()), "sopro_seanet_res128_f32")


def seanet_up128(x: torch.Tensor, w: torch.Tensor, bias: torch.Tensor, out: torch.Tensor, *, B: int, T: int, x_seg_stride: int,
                 out_seg_stride: int, x_off: int = 0, out_off: int = 0, passes: int = 3) -> None:
    """Weight-stationary last transposed convolution of the SEANet decoder (sopro_seanet_up128_f32); offsets in floats."""
    _check(load().sopro_seanet_up128_f32(ptr(x) + 4 * x_off, x_seg_stride, ptr(w), ptr(bias), ptr(out) + 4 * out_off, out_seg_stride, B, T,
                                         passes, _stream()), "sopro_seanet_up128_f32")


def seanet_res128_bf16(h: torch.Tensor, w1: torch.Tensor, b1: torch.Tensor, w2: torch.Tensor, b2: torch.Tensor, out: torch.Tensor, *, B: int,
                       T: int, h_seg_stride: int, out_seg_stride: int) -> None:
    """sopro_seanet_res128_bf16: the fused 128-channel residual block on bf16 rows (strides in bf16 elements)."""
    _check(load().sopro_seanet_res128_bf16(ptr(h, torch.bfloat16), h_seg_stride, ptr(w1), ptr(b1), ptr(w2), ptr(b2), ptr(out, torch.bfloat16),
                                           out_seg_stride, B, T, _stream()), "sopro_seanet_res128_bf16")


def seanet_up128_bf16(x: torch.Tensor, w: torch.Tensor, bias: torch.Tensor, out: torch.Tensor, *, B: int, T: int, x_seg_stride: int,
                      out_seg_stride: int, x_off: int = 0, out_off: int = 0) -> None:
    """sopro_seanet_up128_bf16: the last transposed convolution on bf16 rows (offsets / strides in bf16 elements)."""
    _check(load().sopro_seanet_up128_bf16(ptr(x, torch.bfloat16) + 2 * x_off, x_seg_stride, ptr(w), ptr(bias), ptr(out, torch.bfloat16) + 2 * out_off,
                                          out_seg_stride, B, T, _stream()), "sopro_seanet_up128_bf16")


def seanet_tail_bf16(h: torch.Tensor, w1: torch.Tensor, b1: torch.Tensor, w2: torch.Tensor, b2: torch.Tensor, wf: torch.Tensor,
                     bf: float, wav: torch.Tensor, *, B: int, T: int, h_seg_stride: int, wav_seg_stride: int) -> None:
    """sopro_seanet_tail_bf16: the fused 24 kHz tail on bf16 rows (h_seg_stride in bf16 elements)."""
    _check(load().sopro_seanet_tail_bf16(ptr(h, torch.bfloat16), h_seg_stride, ptr(w1), ptr(b1), ptr(w2), ptr(b2), ptr(wf), bf, ptr(wav),
                                         wav_seg_stride, B, T, _stream()), "sopro_seanet_tail_bf16")


def seanet_uptail(x: torch.Tensor, wu: torch.Tensor, bu: torch.Tensor, w1: torch.Tensor, b1: torch.Tensor, w2: torch.Tensor, b2: torch.Tensor,
                  wf: torch.Tensor, bf: float, wav: torch.Tensor, *, B: int, T: int, x_seg_stride: int, wav_seg_stride: int, x_off: int = 0,
                  passes: int = 3) -> None:
    """The last SEANet level in one kernel (sopro_seanet_uptail_f32 / _bf16 by the dtype of ``x``): the last transposed convolution,
    the last residual block and the last layer; offsets / strides in elements of ``x``."""
    if x.dtype == torch.bfloat16:
        _check(load().sopro_seanet_uptail_bf16(ptr(x, torch.bfloat16) + 2 * x_off, x_seg_stride, ptr(wu), ptr(bu), ptr(w1), ptr(b1), ptr(w2), ptr(b2),
                                               ptr(wf), bf, ptr(wav), wav_seg_stride, B, T, _stream()), "sopro_seanet_uptail_bf16")
    else:
        _check(load().sopro_seanet_uptail_f32(ptr(x) + 4 * x_off, x_seg_stride, ptr(wu), ptr(bu), ptr(w1), ptr(b1), ptr(w2), ptr(b2), ptr(wf), bf,
                                              ptr(wav), wav_seg_stride, B, T, passes, _stream()), "sopro_seanet_uptail_f32")


def fill_u32(t: torch.Tensor, value: int = 0) -> None:
    """Every 32-bit word of a contiguous device tensor <- value, as a kernel of the library (sopro_fill2d_u32) on the current stream."""
    n = t.numel() * t.element_size() // 4
    if not t.is_contiguous() or t.numel() * t.element_size() % 4 or n <= 0:
        raise SoproHipError("fill_u32: a contiguous tensor of whole 32-bit words")
    rows = 1
    while n // rows > (1 << 30):
        rows *= 2
    if n % rows:
        raise SoproHipError("fill_u32: tensor too large for one launch")
    _check(load().sopro_fill2d_u32(t.data_ptr(), n // rows, rows, n // rows, int(value) & 0xFFFFFFFF, _stream()), "sopro_fill2d_u32")


def copy_u32(dst: torch.Tensor, src: torch.Tensor) -> None:
    """dst <- src (contiguous device tensors of the same byte size, whole 32-bit words) as a kernel of the library."""
    n = dst.numel() * dst.element_size() // 4
    if not (dst.is_contiguous() and src.is_contiguous()) or n * 4 != src.numel() * src.element_size() or n <= 0 or n > (1 << 30):
        raise SoproHipError("copy_u32: contiguous tensors of the same size (< 4 GiB)")
    _check(load().sopro_copy2d_u32(dst.data_ptr(), n, src.data_ptr(), n, 1, n, _stream()), "sopro_copy2d_u32")


def set_lds_floor(nbytes: int) -> None:
    """Minimum dynamic-LDS request of the split-bf16 GEMM launches (> 80 KiB: one workgroup per CU); see sopro_set_lds_floor."""
    _check(load().sopro_set_lds_floor(int(nbytes)), "sopro_set_lds_floor")


def ar_init(st: ArState) -> None:
    _check(load().sopro_ar_init(C.byref(st), _stream()), "sopro_ar_init")


def ar_admit(st: ArState, row: int) -> None:
    _check(load().sopro_ar_admit(C.byref(st), int(row), _stream()), "sopro_ar_admit")


def ar_sample(st: ArState, logits: torch.Tensor, ld: int) -> None:
    _check(load().sopro_ar_sample(C.byref(st), ptr(logits), ld, _stream()), "sopro_ar_sample")


def ar_issue_frame(frame: ArFrame) -> None:
    """One autoregressive frame (23 launches) on the current stream: the sequence lives in csrc/ar_frame.hip."""
    _check(load().sopro_ar_issue_frame(C.byref(frame), _stream()), "sopro_ar_issue_frame")


def ar_fold_text(txt: torch.Tensor, nkv_weight: torch.Tensor, kv_w: torch.Tensor, q_wT: torch.Tensor, o_w: torch.Tensor, nkv: torch.Tensor,
                 kvd: torch.Tensor, kp: torch.Tensor, vp: torch.Tensor, *, B: int, S: int, S_cap: int, D: int, H: int, eps: float,
                 out_off: int = 0) -> None:
    """Folded text operands of one AR cross-attention layer (sopro_ar_fold_text); ``out_off``: float offset into kp / vp."""
    _check(load().sopro_ar_fold_text(ptr(txt), ptr(nkv_weight), ptr(kv_w), ptr(q_wT), ptr(o_w), ptr(nkv), ptr(kvd), ptr(kp) + 4 * out_off,
                                     ptr(vp) + 4 * out_off, B, S, S_cap, D, H, eps, _stream()), "sopro_ar_fold_text")


def ar_fold_text_uk(txt: torch.Tensor, nkv_weight: torch.Tensor, kv_w: torch.Tensor, o_w: torch.Tensor, nkv: torch.Tensor, kvd: torch.Tensor,
                    kq: torch.Tensor, vp: torch.Tensor, *, B: int, S: int, S_cap: int, D: int, H: int, eps: float, k_off: int = 0, v_off: int = 0) -> None:
    """Text operands of one AR cross-attention layer with UNFOLDED keys (sopro_ar_fold_text_uk); offsets in floats into kq / vp."""
    _check(load().sopro_ar_fold_text_uk(ptr(txt), ptr(nkv_weight), ptr(kv_w), ptr(o_w), ptr(nkv), ptr(kvd), ptr(kq) + 4 * k_off, ptr(vp) + 4 * v_off,
                                        B, S, S_cap, D, H, eps, _stream()), "sopro_ar_fold_text_uk")


def cvt_f32_bf16(src: torch.Tensor, dst: torch.Tensor, n: Optional[int] = None, dst_off: int = 0) -> None:
    """fp32 -> bf16 image (sopro_cvt_f32_bf16); ``dst_off``: element offset into ``dst``."""
    n = int(src.numel() if n is None else n)
    _check(load().sopro_cvt_f32_bf16(ptr(src), ptr(dst, torch.bfloat16) + 2 * dst_off, n, _stream()), "sopro_cvt_f32_bf16")


def cvt_bf16_f32(src: torch.Tensor, dst: torch.Tensor, n: Optional[int] = None) -> None:
    n = int(src.numel() if n is None else n)
    _check(load().sopro_cvt_bf16_f32(ptr(src, torch.bfloat16), ptr(dst), n, _stream()), "sopro_cvt_bf16_f32")


def set_host_wait(blocking: bool, device=None) -> None:
    """How host threads wait for ``device`` (sopro_set_host_wait -> hipSetDeviceFlags): spin (the runtime's default: one busy core
    per waiting thread) or block on the completion interrupt.  Call it ONCE, AT PROCESS START, before any stream of the device
    exists (before the engine is built): the runtime creates a queue's completion signals for the wait mode in force when the
    queue is made, and a blocking wait on a signal made for spinning never wakes up - measured: hipHostFree hung in a
    process that switched modes after its streams existed (profiles/r04_experiments.md).  A process that runs lanes of a
    pipeline / a serving loop wants True (r04: 0.083 -> 0.040 CPU-s per 24 ms step at unchanged throughput)."""
    with torch.cuda.device(device if device is not None else torch.cuda.current_device()):
        _check(load().sopro_set_host_wait(1 if blocking else 0), "sopro_set_host_wait")


def ar_tile_code(spec: str) -> int:
    """'2x1' -> (mt << 4) | nt of sopro_ar_frame.tile_*"""
    mt, nt = (int(v) for v in spec.lower().split("x"))
    if mt not in (1, 2) or nt not in (1, 2):
        raise ValueError(f"AR tile shape must be 1x1, 1x2, 2x1 or 2x2, got {spec!r}")
    return (mt << 4) | nt



def cond_prepare(engine, ws: torch.Tensor, ids: torch.Tensor, lens: torch.Tensor, ragged: bool, film_mul: torch.Tensor, film_add: torch.Tensor,
                 ref_k, ref_v, kv_bstride: int, kv_index: Optional[torch.Tensor], ref_klens: Optional[torch.Tensor], B: int, S: int, Tar: int,
                 Tr: int, txt_seq: torch.Tensor, txt_pool: torch.Tensor, cond_ar: torch.Tensor) -> None:
    """sopro_cond_prepare: the conditioning launch sequence of B utterances (csrc/stages.hip)."""
    ka = (C.c_void_p * len(ref_k))(*[ptr(t) for t in ref_k])
    va = (C.c_void_p * len(ref_v))(*[ptr(t) for t in ref_v])
    _check(load().sopro_cond_prepare(engine, ptr(ws), ptr(ids, torch.int32), ptr(lens, torch.int32), int(bool(ragged)), ptr(film_mul), ptr(film_add),
                                     ka, va, int(kv_bstride), ptr(kv_index, torch.int32), ptr(ref_klens, torch.int32), B, S, Tar, Tr,
                                     ptr(txt_seq), ptr(txt_pool), ptr(cond_ar), _stream()), "sopro_cond_prepare")


def film_coeffs(engine, sv: torch.Tensor, style: float, n: int, scratch: torch.Tensor, mul: torch.Tensor, add: torch.Tensor) -> None:
    _check(load().sopro_film_coeffs(engine, ptr(sv), float(style), n, ptr(scratch), ptr(mul), ptr(add), _stream()), "sopro_film_coeffs")


def ref_prepare(engine, ws: torch.Tensor, tokens: torch.Tensor, T: int, sv: torch.Tensor, ref_seq: torch.Tensor, kvs) -> None:
    """sopro_ref_prepare: Token2SV, reference encoder and the K | V rows of the reference cross-attention blocks of one voice."""
    if ref_seq is None:  # speaker vector only (SoproTTS.encode_speaker)
        _check(load().sopro_ref_prepare(engine, ptr(ws), ptr(tokens, torch.int32), T, ptr(sv), None, None, _stream()), "sopro_ref_prepare")
        return
    ka = (C.c_void_p * len(kvs))(*[ptr(t) for t in kvs])
    _check(load().sopro_ref_prepare(engine, ptr(ws), ptr(tokens, torch.int32), T, ptr(sv), ptr(ref_seq), ka, _stream()), "sopro_ref_prepare")

class Graph:
    """A recorded launch sequence (hipGraphExec) replayable on any stream."""

    def __init__(self, handle: int, family: str = "ar_step_graph"):
        self.handle = handle
        self.family = family

    def launch(self) -> None:
        e0 = _prof.begin() if _prof is not None else None
        _check(load().sopro_graph_launch(self.handle, _stream()), "sopro_graph_launch")
        if e0 is not None:
            _prof.end(self.family, 0.0, e0)

    def launch_n(self, n: int) -> None:
        """n replays back to back (one call: the interpreter lock is dropped once for all of them)."""
        if _prof is not None:
            for _ in range(int(n)):
                self.launch()
            return
        _check(load().sopro_graph_launch_n(self.handle, _stream(), int(n)), "sopro_graph_launch_n")

    def __del__(self):
        # Never destroy here: the collector may run this INSIDE another recording (any allocation can trigger it), and
        # hipGraphExecDestroy during a stream capture invalidates that capture ("operation failed due to a previous error
        # during capture", seen once in a few suite runs).  The handle is parked and destroyed at the next safe point.
        if self.handle and _dead_graphs is not None:  # (None: module globals already torn down at interpreter exit)
            _dead_graphs.append(self.handle)
            self.handle = 0


_dead_graphs: list = []


def reap_graphs() -> None:
    """Destroy the graphs whose owners are gone; called where no recording can be open (capture_begin takes the lock first)."""
    while _dead_graphs:
        h = _dead_graphs.pop()
        try:
            if _lib is not None:
                _lib.sopro_graph_destroy(h)
        except Exception:
            pass


def profiling() -> bool:
    """True while bench.py's per-launch event timing is attached (recorded graphs would hide the launches from it)."""
    return _prof is not None


class GraphCache:
    """Recorded launch sequences keyed by problem shape, for host code whose launches depend only on that shape.
    A shape is run eagerly the first time (scratch buffers get allocated), recorded the second time, replayed after."""

    def __init__(self, family: str, cap: int = 8):
        self.family, self.cap = family, cap
        self.graphs: dict = {}
        self.seen: dict = {}

    def run(self, key, issue) -> None:
        """``issue()`` enqueues the launches on the current stream; it must not allocate, synchronise or touch the host."""
        if profiling():  # per-launch event timing wants to see the launches one by one
            issue()
            return
        g = self.graphs.get(key)
        if g is not None:
            g.launch()
            return
        n = self.seen.get(key, 0)
        if n == 0 and len(self.seen) >= 8 * max(8, self.cap):  # keys may hold caller pointers (temporary tensors): keep the census bounded
            self.seen.pop(next(iter(self.seen)))
        self.seen[key] = n + 1
        if n == 0:
            issue()
            return
        capture_begin()
        try:
            issue()
        finally:
            g = capture_end()
        g.family = self.family
        if len(self.graphs) >= self.cap:
            self.graphs.pop(next(iter(self.graphs)))
        self.graphs[key] = g
        g.launch()

    def clear(self) -> None:
        self.graphs.clear()
        self.seen.clear()
        reap_parked_graphs()  # the handles were only parked by Graph.__del__: destroy them now if no recording is open


# A recording is thread-local (hipStreamCaptureModeThreadLocal), but the runtime still refused a pinned-host allocation made
# by ANOTHER thread while one was open ("operation not permitted when stream is capturing", once in a few hundred pipelined
# runs, when one lane recorded its launch sequence while another made its first poll buffer).  Recordings and pinned
# allocations are rare and short: they take turns.
_capture_lock = threading.RLock()


def reap_parked_graphs() -> bool:
    """Destroy the parked graph handles NOW if no recording is open in any thread (a warmed-up server may never record again,
    and the handles keep their workspaces' launch descriptors alive: ADVICE r3).  Non-blocking: returns False, leaving them for
    the next capture_begin, when another thread holds the recording lock."""
    if not _dead_graphs:
        return True
    if not _capture_lock.acquire(blocking=False):
        return False
    try:
        if getattr(_capture_depth, "n", 0) == 0:  # (the lock is re-entrant: not inside THIS thread's own recording either)
            reap_graphs()
            return True
        return False
    finally:
        _capture_lock.release()


_capture_depth = threading.local()


_gc_paused = [0, False]  # (open recordings, was the collector enabled) - guarded by _capture_lock


def _gc_pause() -> None:
    """No cyclic garbage collection while a launch sequence is being recorded: a collection runs finalizers of WHATEVER became garbage
    in the process - an engine of an earlier module (sopro_engine_destroy: hipFree), page-locked blocks, torch storages - on the thread
    that happens to allocate next, and a free on the recording thread invalidates the recording ("operation failed due to a previous
    error during capture", seen once in ~20 suite runs, at the first launch of an AR frame recording)."""
    if _gc_paused[0] == 0:
        _gc_paused[1] = gc.isenabled()
        gc.disable()
    _gc_paused[0] += 1


def _gc_resume() -> None:
    _gc_paused[0] = max(0, _gc_paused[0] - 1)
    if _gc_paused[0] == 0 and _gc_paused[1]:
        gc.enable()


def capture_begin() -> None:
    _capture_lock.acquire()
    try:
        if getattr(_capture_depth, "n", 0) == 0:
            reap_graphs()
        _gc_pause()
        try:
            _check(load().sopro_capture_begin(_stream()), "sopro_capture_begin")
        except BaseException:
            _gc_resume()
            raise
        _capture_depth.n = getattr(_capture_depth, "n", 0) + 1
    except BaseException:
        _capture_lock.release()
        raise


def capture_end() -> Graph:
    out = _p()
    try:
        _check(load().sopro_capture_end(_stream(), C.byref(out)), "sopro_capture_end")
    finally:
        _capture_depth.n = max(0, getattr(_capture_depth, "n", 0) - 1)
        _gc_resume()
        _capture_lock.release()
    return Graph(out.value)


class HostMirror:
    """``n`` int32 of page-locked host memory that a host loop polls: ``copy_from(t)`` queues the device -> host copy behind
    the launches issued so far on the current stream (record an event after it and wait for that event before reading
    ``values()``).  Outside torch's pinned-memory cache on purpose (see sopro_host_alloc)."""

    def __init__(self, n: int):
        self.n = int(n)
        out = _p()
        with _capture_lock:  # no page-locked allocation while another thread records a launch sequence
            _check(load().sopro_host_alloc(4 * self.n, C.byref(out)), "sopro_host_alloc")
        self.ptr = out.value
        self._view = (C.c_int32 * self.n).from_address(self.ptr)
        self._keep = None

    def copy_from(self, t: torch.Tensor) -> None:
        """Queue device -> host behind the current stream's launches.  A KERNEL of the library writes the page-locked words
        (sopro_copy2d_u32): the timed path holds no runtime copy - no `__amd_rocclr_copyBuffer`, no DMA packet - at all (round 5)."""
        if t.dtype not in (torch.int32, torch.float32) or not t.is_contiguous() or t.numel() != self.n or not t.is_cuda:
            raise SoproHipError(f"HostMirror.copy_from: expected a contiguous device int32 / float32 tensor of {self.n} elements")
        self._keep = t  # the source of a queued copy stays alive
        _check(load().sopro_copy2d_u32(self.ptr, self.n, t.data_ptr(), self.n, 1, self.n, _stream()), "sopro_copy2d_u32")

    def copy_to(self, t: torch.Tensor) -> None:
        """Queue host -> device: the kernel reads the page-locked words when it RUNS, so the host must leave them alone until the
        stream has passed this point (the users write a block, queue the launches that read it and synchronise before the next write)."""
        m = t.numel()  # (a block may be larger than what this call sends: the first m words go)
        if t.dtype not in (torch.int32, torch.float32) or not t.is_contiguous() or not (0 < m <= self.n) or not t.is_cuda or not self.ptr:
            raise SoproHipError(f"HostMirror.copy_to: expected a contiguous device int32 / float32 tensor of at most {self.n} elements")
        _check(load().sopro_copy2d_u32(t.data_ptr(), m, self.ptr, m, 1, m, _stream()), "sopro_copy2d_u32")

    def values(self) -> list:
        return list(self._view)

    def array(self):
        """The words as a writable numpy int32 view (``.view(np.float32)`` for float parameters)."""
        import numpy as np

        return np.ctypeslib.as_array(self._view)

    def free(self) -> None:
        """Give the page-locked words back NOW.  The caller has made sure no queued kernel and no recorded launch sequence still
        holds the address.  Under the recording lock, like the allocation: the runtime refuses page-locked (de)allocations made
        while another thread records."""
        p, self.ptr = getattr(self, "ptr", None), None
        if p and _lib is not None:
            self._view = None
            with _capture_lock:
                _lib.sopro_host_free(p)

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass


def cu_range_stream(first_cu: int, n_cus: int, device: Optional[torch.device] = None) -> "torch.cuda.Stream":
    """A torch-visible stream whose kernels are confined to CUs [first_cu, first_cu + n_cus).
    The owner must hand it back with ``destroy_stream`` (torch does not own external streams)."""
    out = _p()
    with torch.cuda.device(device if device is not None else torch.cuda.current_device()):  # the C side creates it on the current device
        _check(load().sopro_stream_create_cu_range(first_cu, n_cus, C.byref(out)), "sopro_stream_create_cu_range")
    return torch.cuda.ExternalStream(out.value, device=device)


def cu_mask_stream(cus, device: Optional[torch.device] = None) -> "torch.cuda.Stream":
    """A torch-visible stream confined to the CUs whose mask bits are in ``cus`` (see sopro_stream_create_cu_mask for what
    the bits mean on gfx950).  Hand it back with ``destroy_stream``."""
    cus = sorted(set(int(c) for c in cus))
    words = (max(cus) >> 5) + 1
    arr = (C.c_uint32 * words)()
    for c in cus:
        arr[c >> 5] |= 1 << (c & 31)
    out = _p()
    with torch.cuda.device(device if device is not None else torch.cuda.current_device()):
        _check(load().sopro_stream_create_cu_mask(arr, words, C.byref(out)), "sopro_stream_create_cu_mask")
    return torch.cuda.ExternalStream(out.value, device=device)


def destroy_stream(s: "torch.cuda.Stream") -> None:
    _splitk_pool.pop(s.cuda_stream, None)
    _check(load().sopro_stream_destroy(s.cuda_stream), "sopro_stream_destroy")


def device_info(device: int = 0) -> dict:
    arr = (C.c_int * 4)()
    _check(load().sopro_device_info(device, arr), "sopro_device_info")
    return {"cus": arr[0], "lds_per_block": arr[1], "clock_khz": arr[2], "gfx": arr[3]}
