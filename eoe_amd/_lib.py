"""ctypes binding of libeoe_hip.so (the C ABI in include/eoe_hip.h).  There is NO fallback: if the library is
missing and cannot be built, importing this module raises."""
import ctypes as C
import os
import re

# PyTorch ships its own libamdhip64.so; libeoe_hip.so is linked against the system one (same SONAME).  Whichever is loaded
# first serves both -- and with the system runtime loaded first the process ends up with TWO HIP runtimes, torch on its own
# and this library on one that reports "no ROCm-capable device".  Import torch first so that its runtime is the one in the
# process (the tensors, streams and events handed to the C ABI belong to it).
import torch  # noqa: F401  (must precede the CDLL below)

from . import _build

ABI_VERSION = 5
EOE_F16, EOE_BF16, EOE_F32 = 1, 2, 3
EOE_RESIZE_BILINEAR, EOE_RESIZE_BICUBIC = 2, 3
EOE_COMM_I64, EOE_COMM_ID_BYTES, EOE_COMM_ALGO_RING, EOE_COMM_ALGO_RS_AG = 8, 128, 0, 1
EPI_NONE, EPI_GELU, EPI_RESIDUAL, EPI_GELU_BWD = 0, 1, 2, 3
ADAM_CHUNK, ADAM_GROUPS = 8192, 4
CHUNK_FP16 = 0x100                 # eoe_adam_chunk.group flag (EOE_CHUNK_FP16): fp16-weights update in eoe_sgd_multi
EOE_Y16 = 0x100             # OR-ed into the dtype argument of eoe_bn_act_*: y is the 16-bit copy (include/eoe_hip.h)

_vp, _i32, _i64, _f32, _sz = C.c_void_p, C.c_int32, C.c_int64, C.c_float, C.c_size_t


class ConvGeometry(C.Structure):
    _fields_ = [("n", _i32), ("H", _i32), ("W", _i32), ("C", _i32), ("kh", _i32), ("kw", _i32), ("stride", _i32), ("pad", _i32),
                ("Ho", _i32), ("Wo", _i32)]


class GemmArgs(C.Structure):
    _fields_ = [("A", _vp), ("B", _vp), ("C", _vp), ("bias", _vp), ("aux", _vp), ("aux_out", _vp), ("colsum", _vp),
                ("M", _i32), ("N", _i32), ("K", _i32),
                ("lda", _i32), ("ldb", _i32), ("ldc", _i32), ("ldaux", _i32),
                ("dtype", _i32), ("epilogue", _i32), ("out_f32", _i32), ("accumulate", _i32), ("alpha", _f32),
                ("workspace", _vp), ("workspace_bytes", _i64), ("gather", _i32), ("geo", ConvGeometry), ("colstats", _i32),
                ("unpack_dw", _i32), ("split_k", _i32), ("sk_workspace", _vp), ("sk_workspace_bytes", _i64)]


class AdamChunk(C.Structure):
    _fields_ = [("p_off", _i64), ("g_off", _i64), ("m_off", _i64), ("v_off", _i64), ("n", _i32), ("group", _i32)]


class AdamTile(C.Structure):
    _fields_ = [("p_off", _i64), ("g_off", _i64), ("m_off", _i64), ("v_off", _i64), ("d16", _vp), ("d16_t", _vp), ("rows", _i32),
                ("cols", _i32), ("tile", _i32), ("group", _i32)]


class AdamScalars(C.Structure):
    _fields_ = [("step_size", _f32 * ADAM_GROUPS), ("bc2_sqrt", _f32 * ADAM_GROUPS), ("grad_scale_inv", _f32)]


class CGateArgs(C.Structure):
    _fields_ = [("x", _vp), ("out", _vp), ("w1", _vp), ("b1", _vp), ("w2", _vp), ("b2", _vp), ("pooled", _vp), ("argmax", _vp),
                ("hidden", _vp), ("scale", _vp), ("n", _i32), ("HW", _i32), ("C", _i32), ("Ch", _i32)]


class CGateBwdArgs(C.Structure):
    _fields_ = [("f", CGateArgs), ("dout", _vp), ("dx", _vp), ("dscale", _vp), ("dpooled", _vp), ("dhidden", _vp),
                ("dw1", _vp), ("db1", _vp), ("dw2", _vp), ("db2", _vp)]


class SGateArgs(C.Structure):
    _fields_ = [("x", _vp), ("out", _vp), ("w", _vp), ("gamma", _vp), ("beta", _vp), ("running_mean", _vp), ("running_var", _vp),
                ("num_batches_tracked", _vp), ("comp", _vp), ("argmax", _vp), ("z", _vp), ("stats", _vp), ("scale", _vp),
                ("sums", _vp), ("n", _i32), ("H", _i32), ("W", _i32), ("C", _i32), ("eps", _f32), ("momentum", _f32),
                ("training", _i32), ("res", _vp), ("out16", _vp), ("dtype", _i32)]


class SGateBwdArgs(C.Structure):
    _fields_ = [("f", SGateArgs), ("dout", _vp), ("dx", _vp), ("dscale", _vp), ("dcomp", _vp), ("red", _vp), ("dw", _vp),
                ("dgamma", _vp), ("dbeta", _vp), ("wpart", _vp)]


class ConvPackJob(C.Structure):
    _fields_ = [("w", _vp), ("w16", _vp), ("w16t", _vp), ("w16d", _vp), ("cout", _i32), ("cin", _i32), ("cpad", _i32), ("kh", _i32),
                ("kw", _i32), ("Kp", _i32)]


class CastJob(C.Structure):
    _fields_ = [("src", _vp), ("dst", _vp), ("dst_t", _vp), ("rows", _i32), ("cols", _i32)]


class ProfEntry(C.Structure):
    _fields_ = [("name", C.c_char * 32), ("launches", _i64), ("total_ms", C.c_double), ("flops", C.c_double),
                ("bytes", C.c_double)]


class VitBlockFwdArgs(C.Structure):
    _fields_ = [("n", _i32), ("L", _i32), ("D", _i32), ("heads", _i32), ("dtype", _i32), ("eps", _f32),
                ("ln1_g", _vp), ("ln1_b", _vp), ("ln2_g", _vp), ("ln2_b", _vp),
                ("b_in", _vp), ("b_out", _vp), ("b_fc", _vp), ("b_proj", _vp),
                ("w_in", _vp), ("w_out", _vp), ("w_fc", _vp), ("w_proj", _vp),
                ("w_in_t", _vp), ("w_out_t", _vp), ("w_fc_t", _vp), ("w_proj_t", _vp),
                ("x_in", _vp), ("x_mid", _vp), ("x_out", _vp),
                ("xn1", _vp), ("qkv", _vp), ("att", _vp), ("xn2", _vp), ("hpre", _vp), ("hact", _vp),
                ("stats1", _vp), ("stats2", _vp), ("cls_only", _i32), ("nt_sk_workspace", _vp), ("nt_sk_workspace_bytes", _i64)]


class RedJob(C.Structure):
    _fields_ = [("part", _vp), ("R", _i32), ("N", _i32), ("seg", _i32), ("blocked", _i32), ("out", _vp * 3)]


RED_TABLE_MAX = 128


class RedTable(C.Structure):
    _fields_ = [("job", RedJob * RED_TABLE_MAX), ("count", _i32), ("overwrite", _i32)]


class VitBlockBwdArgs(C.Structure):
    _fields_ = [("f", VitBlockFwdArgs), ("dx_out", _vp), ("dx_in", _vp),
                ("g_ln1_g", _vp), ("g_ln1_b", _vp), ("g_ln2_g", _vp), ("g_ln2_b", _vp),
                ("g_b_in", _vp), ("g_b_out", _vp), ("g_b_fc", _vp), ("g_b_proj", _vp),
                ("g_w_in", _vp), ("g_w_out", _vp), ("g_w_fc", _vp), ("g_w_proj", _vp),
                ("accumulate", _i32),
                ("d16_a", _vp), ("d16_b", _vp), ("d16_c", _vp), ("dh", _vp), ("dqkv", _vp), ("dx_mid", _vp), ("red_scratch", _vp),
                ("tn_workspace", _vp), ("tn_workspace_bytes", _i64),
                ("next_d16", _vp), ("in_d16", _vp), ("in_red_scratch", _vp), ("async_wgrad", _i32), ("red_table", C.POINTER(RedTable))]


# name -> argtypes (restype is int unless listed in _RESTYPES); must match include/eoe_hip.h
# eoe_allreduce_fn of include/eoe_hip.h (synchronised BatchNorm hook)
ALLREDUCE_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.c_int64, C.c_int, C.c_void_p)

SIGNATURES = {
    "eoe_abi_version": [],
    "eoe_struct_size": [C.c_int],
    "eoe_last_error": [],
    "eoe_vit_side_join": [_vp],
    "eoe_red_table_flush": [C.POINTER(RedTable), _vp],
    "eoe_gemm_nt": [C.POINTER(GemmArgs), _vp],
    "eoe_gemm_tn": [C.POINTER(GemmArgs), _vp],
    "eoe_gemm_tn_grouped": [C.POINTER(GemmArgs), C.c_int, _vp],
    "eoe_cast_transpose": [_vp, _vp, _vp, C.c_int, C.c_int, C.c_int, _vp],
    "eoe_cast_transpose_multi": [_vp, C.c_int, C.c_int, _vp],
    "eoe_conv_pack_weight_multi": [_vp, C.c_int, C.c_int, _vp],
    "eoe_patchify": [_vp, _vp, _vp, _vp, C.c_int, C.c_int, C.c_int, C.c_int, _vp],
    "eoe_embed_lnpre_fwd": [_vp] * 8 + [C.c_int, C.c_int, C.c_int, _f32, _vp],
    "eoe_embed_lnpre_bwd": [_vp] * 10 + [C.c_int, C.c_int, C.c_int, C.c_int, _vp],
    "eoe_layernorm_fwd": [_vp, C.c_int, _vp, _vp, _vp, _vp, C.c_int, C.c_int, _f32, C.c_int, C.c_int, _vp],
    "eoe_layernorm_bwd": [_vp, C.c_int, _vp, C.c_int, _vp, _vp, _vp, _vp, C.c_int, _vp, _vp, _vp, _vp, _vp, C.c_int,
                          C.c_int, C.c_int, _vp],
    "eoe_linear_small_fwd": [_vp, _vp, _vp, _vp, C.c_int, C.c_int, C.c_int, _vp],
    "eoe_linear_small_bwd": [_vp, _vp, _vp, _vp, _vp, _vp, C.c_int, C.c_int, C.c_int, C.c_int, _vp],
    "eoe_zero_multi": [C.POINTER(_vp), C.POINTER(C.c_int), C.c_int, _vp],
    "eoe_cast_colsum": [_vp, _vp, _vp, _vp, C.c_int, C.c_int, C.c_int, C.c_int, _vp],
    "eoe_colsum": [_vp, C.c_int, _vp, C.c_int, C.c_int, C.c_int, C.c_int, _vp],
    "eoe_colsum_det": [_vp, C.c_int, _vp, _vp, C.c_int, C.c_int, C.c_int, _vp],
    "eoe_cast": [_vp, _vp, _sz, C.c_int, _vp],
    "eoe_attn_fwd": [_vp, _vp, C.c_int, C.c_int, C.c_int, C.c_int, _vp],
    "eoe_attn_bwd": [_vp, _vp, _vp, _vp, _vp, C.c_int, C.c_int, C.c_int, C.c_int, _vp],
    "eoe_hsc_fwd": [_vp, _vp, _i64, _vp, _vp, _vp, _vp, C.c_int, C.c_int, _f32, _vp],
    "eoe_hsc_bwd": [_vp, _vp, _i64, _vp, _vp, _vp, C.c_int, C.c_int, _f32, C.c_int, _vp],
    "eoe_hsc_score": [_vp, _vp, C.c_int, C.c_int, _vp],
    "eoe_bce_fwd": [_vp, _vp, _i64, _vp, _vp, _vp, C.c_int, _f32, _vp],
    "eoe_bce_bwd": [_vp, _vp, _vp, _vp, C.c_int, _f32, _vp],
    "eoe_adam_tiles": [_vp, _vp, _vp, _vp, _vp, C.c_int, C.POINTER(AdamScalars), _f32, _f32, _f32, _f32, C.c_int, _vp, _vp],
    "eoe_adam_multi": [_vp, _vp, _vp, _vp, _vp, C.c_int, C.POINTER(AdamScalars), _f32, _f32, _f32, _f32, _vp, C.c_int,
                       _vp, _vp],
    "eoe_grads_nonfinite": [_vp, _vp, C.c_int, _vp, C.c_int, C.c_int, _vp],
    "eoe_vit_block_fwd": [C.POINTER(VitBlockFwdArgs), _vp],
    "eoe_vit_block_bwd": [C.POINTER(VitBlockBwdArgs), _vp],
    "eoe_im2col": [_vp, C.c_int, _vp, _vp, _vp] + [C.c_int] * 10 + [_vp],
    "eoe_col2im": [_vp, _vp] + [C.c_int] * 11 + [_vp],
    "eoe_conv_pack_weight": [_vp, _vp, _vp, _vp] + [C.c_int] * 7 + [_vp],
    "eoe_conv_unpack_wgrad": [_vp, _vp] + [C.c_int] * 8 + [_vp],
    "eoe_stem_pack_image": [_vp, _vp, _vp, _vp] + [C.c_int] * 8 + [_vp],
    "eoe_stem_pack_weight": [_vp, _vp] + [C.c_int] * 4 + [_vp],
    "eoe_stem_unpack_wgrad": [_vp, _vp] + [C.c_int] * 3 + [_vp],
    "eoe_bn_stats": [_vp, _vp, _vp, _vp, _vp, _vp, C.c_int, C.c_int, _f32, _f32, C.c_int, _vp],
    "eoe_bn_stats_partials": [_vp, C.c_int, _vp, _vp, _vp, _vp, _vp, C.c_int, C.c_int, _f32, _f32, _vp],
    "eoe_bn_act_pool_fwd": [_vp, _vp, _vp, _vp, _vp, _vp, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, _f32,
                            C.c_int, _vp],
    "eoe_colsum_f32": [_vp, _vp, _vp, C.c_int, C.c_int, C.c_int, _vp],
    "eoe_bn_act_pool_bwd": [_vp, _vp, _vp, _vp, _vp, _vp, _vp, C.c_int, _vp, _vp, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int,
                            C.c_int, C.c_int, _f32, C.c_int, _vp],
    "eoe_bn_act_maxpool_fwd": [_vp] * 7 + [C.c_int] * 7 + [_f32, C.c_int, _vp],
    "eoe_bn_act_maxpool_bwd": [_vp] * 10 + [C.c_int] * 8 + [_f32, C.c_int, _vp],
    "eoe_pack_image_nhwc4": [_vp, _vp, _vp, _vp, C.c_int, C.c_int, C.c_int, _vp],
    "eoe_conv_f32_pack_weights": [_vp, _vp, _vp, C.c_int, C.c_int, C.c_int, C.c_int, _vp],
    "eoe_conv_f32_fwd": [_vp, C.c_int, _vp, _vp, _vp, _vp, _vp, C.POINTER(ConvGeometry), C.c_int, _vp, C.c_size_t, _vp, _vp],
    "eoe_conv_f32_dgrad": [_vp, _vp, _vp, C.POINTER(ConvGeometry), C.c_int, C.c_int, _vp, C.c_size_t, _vp, _vp],
    "eoe_conv_f32_wgrad_workspace": [C.POINTER(ConvGeometry), C.c_int],
    "eoe_conv_f32_wgrad": [_vp, C.c_int, _vp, _vp, _vp, _vp, C.POINTER(ConvGeometry), C.c_int, _vp, _sz, _vp],
    "eoe_maxpool_fwd": [_vp, _vp, _vp, _vp] + [C.c_int] * 8 + [_vp],
    "eoe_maxpool_bwd": [_vp, _vp, _vp] + [C.c_int] * 7 + [_vp],
    "eoe_cgate_fwd": [C.POINTER(CGateArgs), _vp],
    "eoe_cgate_bwd": [C.POINTER(CGateBwdArgs), _vp],
    "eoe_sgate_fwd": [C.POINTER(SGateArgs), _vp],
    "eoe_sgate_bwd": [C.POINTER(SGateBwdArgs), _vp],
    "eoe_cbam_junction_fwd": [C.POINTER(CGateArgs), C.POINTER(SGateArgs), _vp],
    "eoe_cbam_junction_bwd": [C.POINTER(CGateBwdArgs), C.POINTER(SGateBwdArgs), _vp, _vp, _vp, _vp],
    "eoe_add_relu_fwd": [_vp, _vp, _vp, _vp, C.c_int, _i64, _vp],
    "eoe_relu_bwd": [_vp, _vp, _vp, _i64, _vp],
    "eoe_avgpool_fwd": [_vp, _vp, _vp, C.c_int, C.c_int, C.c_int, _vp],
    "eoe_avgpool_bwd": [_vp, _vp, C.c_int, C.c_int, C.c_int, _vp],
    "eoe_auc_ap": [_vp, _vp, _i64, _vp, _vp, C.c_int, _vp],
    "eoe_clip_fwd": [_vp, _vp, _vp, _i64, C.c_int, _vp, _vp, _vp, C.c_int, C.c_int, C.c_int, _f32, _vp],
    "eoe_clip_bwd": [_vp, _vp, _vp, _i64, C.c_int, _vp, _vp, C.c_int, C.c_int, C.c_int, _f32, _vp],
    "eoe_clip_score": [_vp, _vp, _vp, C.c_int, C.c_int, C.c_int, _vp],
    "eoe_sgd_multi": [_vp, _vp, _vp, _vp, C.c_int, _f32, _f32, _f32, C.c_int, _f32, _vp, _vp],
    "eoe_dsad_fwd": [_vp, _vp, _i64, _vp, _vp, C.c_int, C.c_int, _f32, _vp],
    "eoe_dsad_bwd": [_vp, _vp, _i64, _vp, _vp, C.c_int, C.c_int, _f32, _vp],
    "eoe_dsvdd_fwd": [_vp, _vp, _vp, _vp, C.c_int, C.c_int, _f32, _vp],
    "eoe_dsvdd_bwd": [_vp, _vp, _vp, _vp, C.c_int, C.c_int, _f32, _vp],
    "eoe_focal_fwd": [_vp, _vp, _i64, _vp, _vp, _vp, C.c_int, _f32, _f32, _f32, _vp],
    "eoe_focal_bwd": [_vp, _vp, _vp, _vp, C.c_int, _f32, _f32, _f32, _vp],
    "eoe_augment_batch": [_vp, _i64, C.c_int, C.c_int, _vp, _vp, _vp, _vp, C.c_int, C.c_int, C.c_int, C.c_int, _f32, C.c_uint64, _vp],
    "eoe_resize_coeffs": [C.c_int, C.c_int, C.c_int, _vp, _vp, C.c_int, C.POINTER(C.c_int)],
    "eoe_resize_pass_u8": [_vp, _vp, _vp, _vp, C.c_int, _i64, C.c_int, C.c_int, C.c_int, _vp],
    "eoe_color_jitter_u8": [_vp, _i64, _vp, _vp, _vp, _vp, _vp, C.c_int, C.c_int, C.c_int, _vp],
    "eoe_comm_unique_id": [_vp],
    "eoe_comm_init": [_vp, C.c_int, C.c_int, C.c_int, C.POINTER(_vp)],
    "eoe_comm_destroy": [_vp],
    "eoe_comm_info": [_vp, C.POINTER(C.c_int), C.POINTER(C.c_int)],
    "eoe_comm_allreduce_sum_async": [_vp, _vp, _i64, C.c_int, C.c_int, _vp],
    "eoe_comm_allgather_async": [_vp, _vp, _vp, _i64, C.c_int, _vp],
    "eoe_comm_join": [_vp, _vp],
    "eoe_comm_sync_bn": [_vp, C.c_int, _vp],
    "eoe_set_bn_sync": [_vp, _vp],
    "eoe_probe_mfma_f16": [_vp, C.c_int, C.c_int, _vp],
    "eoe_probe_copy": [_vp, _vp, _i64, _vp],
    "eoe_prof_enable": [C.c_int],
    "eoe_set_option": [C.c_char_p, C.c_int],
    "eoe_get_option": [C.c_char_p, C.POINTER(C.c_int)],
    "eoe_debug_gemm_stamps": [_vp, C.c_int],
    "eoe_prof_collect": [C.POINTER(ProfEntry), C.c_int, C.POINTER(C.c_int)],
}
_RESTYPES = {"eoe_last_error": C.c_char_p, "eoe_conv_f32_wgrad_workspace": _sz}


def header_symbols():
    """names of every function declared in include/eoe_hip.h (used by the CPU test of the export list)"""
    hdr = os.path.join(os.path.dirname(_build.HERE), "include", "eoe_hip.h")
    txt = open(hdr).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b(eoe_[a-z0-9_]+)\s*\(", txt)))


# the argument structs mirrored above, by their eoe_struct_size index (include/eoe_hip.h)
_STRUCTS = {0: GemmArgs, 1: ConvGeometry, 2: AdamChunk, 3: AdamScalars, 4: VitBlockFwdArgs, 5: VitBlockBwdArgs, 6: CGateArgs,
            7: CGateBwdArgs, 8: SGateArgs, 9: SGateBwdArgs, 10: AdamTile, 11: RedTable}


def _load():
    if _build.needs_build():
        if _build.hipcc_path() is None:
            # a library that does not match the sources must not be loaded silently: a struct-only change of the C ABI would
            # corrupt memory without any symbol going missing
            what = "is missing" if not os.path.exists(_build.LIB) else "does not match the sources (content stamp)"
            raise ImportError(f"eoe_amd: libeoe_hip.so {what} and hipcc is not available to build it; there is no CPU fallback")
        _build.build(verbose=False)
    lib = C.CDLL(_build.LIB)
    for name, argtypes in SIGNATURES.items():
        fn = getattr(lib, name)            # AttributeError here = the library does not export the ABI
        fn.argtypes = argtypes
        fn.restype = _RESTYPES.get(name, C.c_int)
    v = lib.eoe_abi_version()
    if v != ABI_VERSION:
        raise ImportError(f"eoe_amd: libeoe_hip.so has ABI version {v}, expected {ABI_VERSION}")
    for idx, cls in _STRUCTS.items():
        if lib.eoe_struct_size(idx) != C.sizeof(cls):
            raise ImportError(f"eoe_amd: {cls.__name__} is {C.sizeof(cls)} bytes here but {lib.eoe_struct_size(idx)} in libeoe_hip.so "
                              "(the ctypes mirror and include/eoe_hip.h disagree)")
    return lib


def get_option(name: str) -> int:
    v = C.c_int(0)
    check(lib.eoe_get_option(name.encode(), C.byref(v)), "eoe_get_option")
    return v.value


def set_option(name: str, value: int) -> int:
    """sets a tuning switch and returns the value it had"""
    old = get_option(name)
    check(lib.eoe_set_option(name.encode(), int(value)), "eoe_set_option")
    return old


lib = _load()


class EoeError(RuntimeError):
    pass


def check(rc: int, what: str = ""):
    if rc != 0:
        msg = lib.eoe_last_error()
        raise EoeError(f"{what}: error {rc}: {msg.decode() if msg else ''}")


def prof_enable(on: bool):
    check(lib.eoe_prof_enable(1 if on else 0), "eoe_prof_enable")


def prof_collect():
    """{kernel name: dict(launches, total_ms, flops, bytes)} for everything recorded since prof_enable(True)"""
    buf = (ProfEntry * 64)()
    n = C.c_int(0)
    check(lib.eoe_prof_collect(buf, 64, C.byref(n)), "eoe_prof_collect")
    return {buf[i].name.decode(): dict(launches=buf[i].launches, total_ms=buf[i].total_ms, flops=buf[i].flops,
                                       bytes=buf[i].bytes) for i in range(n.value)}



