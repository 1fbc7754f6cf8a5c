"""ctypes binding of libgomatching_hip.so (the C ABI in include/gomatching_hip.h).

The product path has NO fallback: if the shared library is missing or cannot be loaded, importing
the ops raises.  torch is used for device memory and streams only.
"""
import ctypes
import os
from ctypes import c_int, c_long, c_float, c_void_p

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("GOM_LIB_PATH") or os.path.join(HERE, "libgomatching_hip.so")   # (override: same-box A/B of two builds)

GOM_OK = 0
_ERR = {1: "GOM_ERR_INVALID_ARG (violated precondition)", 2: "GOM_ERR_UNSUPPORTED"}

P = c_void_p
I = c_int
L = c_long
F = c_float

# name -> (restype, argtypes); must list every symbol declared in include/gomatching_hip.h
SIGNATURES = {
    "gom_abi_version": (I, []),
    "gom_built_for_arch": (ctypes.c_char_p, []),
    "gom_ms_deform_attn_forward": (I, [P, P, P, P, P, P, I, I, I, I, I, I, I, P]),
    "gom_ms_deform_attn_forward_any": (I, [I, P, P, P, P, P, P, I, I, I, I, I, I, I, P]),
    "gom_ms_deform_attn_backward": (I, [I, P, P, P, P, P, P, P, P, P, I, I, I, I, I, I, I, P]),
    "gom_ms_deform_attn_forward_strided": (I, [P, L, I, P, P, P, P, P, I, I, P]),
    "gom_msda_set_lane_distributed": (I, [I]),
    "gom_msda_fused_forward": (I, [P, I, P, P, L, I, P, P, P, I, I, P]),
    "gom_msda_fused_forward_encoder": (I, [P, I, P, P, L, I, P, P, P, I, I, I, I, I, I, P]),
    "gom_msda_set_window": (I, [I]),
    "gom_msda_window_count_fallbacks": (I, [P]),
    "gom_msda_prepare": (I, [P, I, P, I, P, P, P, L, P]),
    "gom_gemm_f32": (I, [P, P, P, I, P, I, P, P, P, I, I, P, I, I, I, I, P]),
    "gom_gemm_splitk_workspace_bytes": (L, [I, I, I]),
    "gom_gemm_f32_splitk": (I, [P, P, I, P, I, P, P, P, I, I, P, I, I, I, I, P, L, P]),
    "gom_conv2d_nhwc_f32": (I, [P, P, P, P, P, I, P, I, I, I, I, I, I, I, I, I, P]),
    "gom_split_bf16x3": (I, [P, I, I, I, P, I, P]),
    "gom_gemm_f32_bf16x6": (I, [P, P, I, P, L, I, P, P, P, I, I, I, P, I, I, I, I, P]),
    "gom_conv2d_nhwc_f32_bf16x6": (I, [P, P, L, I, P, P, P, I, P, I, I, I, I, I, I, I, I, I, P]),
    "gom_gemm_small_f32": (I, [P, P, I, P, I, P, P, P, I, I, P, I, I, I, I, P]),
    "gom_conv_bf16x6_splits": (I, [I, I, I]),
    "gom_conv2d_nhwc_f32_bf16x6_splitk": (I, [P, P, L, I, P, P, P, I, P, I, I, I, I, I, I, I, I, I, P, L, I, P]),
    "gom_split_f16x2": (I, [P, I, I, I, P, I, P, P]),
    "gom_gemm_f32_f16x3": (I, [P, P, I, P, L, I, P, P, P, P, I, I, I, P, I, I, I, I, P, P]),
    "gom_gemm_f32_f16x3_rp": (I, [P, P, I, P, L, I, P, P, P, P, I, I, I, I, P, I, I, I, I, P, P]),
    "gom_conv2d_nhwc_f32_f16x3": (I, [P, P, L, I, P, P, P, P, I, P, I, I, I, I, I, I, I, I, I, P, L, I, P, P]),
    "gom_conv3x3_patch_supported": (I, [I, I]),
    "gom_conv3x3_patch_image_bytes": (L, [I, I]),
    "gom_conv3x3_patch_image": (I, [P, L, I, I, I, P, L, P]),
    "gom_conv3x3_patch_f32_f16x3": (I, [P, P, P, P, P, I, P, I, I, I, I, I, P, P]),
    "gom_bneck_image_bytes": (L, [I, I, I]),
    "gom_bneck_image": (I, [P, L, I, P, P, P, P, L, I, I, I, I, P, L, P]),
    "gom_bneck_f32": (I, [P, I, P, P, I, P, P, P, I, P, I, I, I, I, I, P, P]),
    "gom_bneck2_image_bytes": (L, [I, I, I]),
    "gom_bneck2_image": (I, [P, L, I, P, P, P, P, L, I, I, I, I, P, L, P]),
    "gom_bneck2_f32": (I, [P, I, P, P, I, P, P, P, I, P, I, I, I, I, I, P, P]),
    "gom_dec_attn_image_bytes": (L, [I, I]),
    "gom_dec_attn_image": (I, [P, L, I, P, P, P, L, I, P, P, P, P, I, P, L, P]),
    "gom_dec_attn_f32": (I, [P, I, P, I, P, F, P, I, I, I, I, I, P, P]),
    "gom_dec_attn_raw_image_bytes": (L, []),
    "gom_dec_attn_raw_image": (I, [P, L, I, P, P, P, L, P]),
    "gom_dec_attn_raw_f32": (I, [P, I, P, F, P, I, P, I, P, I, I, I, I, P, P]),
    "gom_dec_attn2_image_bytes": (L, [I, I]),
    "gom_dec_attn2_image": (I, [P, L, I, P, P, P, L, I, P, P, P, P, I, P, L, P]),
    "gom_dec_attn2_f32": (I, [P, I, P, I, P, F, P, I, I, I, I, I, P, P]),
    "gom_dec_attn2_raw_image_bytes": (L, []),
    "gom_dec_attn2_raw_image": (I, [P, L, I, P, P, P, L, P]),
    "gom_dec_attn2_raw_f32": (I, [P, I, P, F, P, I, P, I, P, I, I, I, I, P, P]),
    "gom_dec_inter_heads_f32": (I, [P, I, P, P, I, I, I, I, P, P]),
    "gom_proj_ln_image_bytes": (L, [I, I]),
    "gom_proj_ln_image": (I, [P, L, I, I, I, P, L, P]),
    "gom_proj_ln_f32": (I, [P, I, P, P, P, P, I, P, P, F, P, I, I, P, P]),
    "gom_proj_ln_dot_f32": (I, [P, I, P, P, P, P, P, F, P, F, P, I, P, P]),
    "gom_gemm_k256_image_bytes": (L, [I, I]),
    "gom_gemm_k256_image": (I, [P, L, I, P, P, I, I, P, L, P]),
    "gom_gemm_k256_f32": (I, [P, P, I, P, P, I, I, I, P, I, I, I, I, I, P, P]),
    "gom_gemm_k256_rp_f32": (I, [P, P, I, P, P, I, I, I, I, P, I, I, I, I, I, P, P]),
    "gom_gemm_k256_set_lines": (None, [I]),
    "gom_gemm_k256_set_interleave": (None, [I]),
    "gom_stem_conv_pool_f32": (I, [P, P, L, I, P, P, P, P, I, I, I, P, P]),
    "gom_ffn_fused_image_bytes": (L, [I, I]),
    "gom_ffn_fused_image": (I, [P, L, I, P, P, P, L, I, I, I, P, L, P]),
    "gom_ffn_set_half_tail": (I, [I]),
    "gom_ffn_set_stream_cus": (I, [I]),
    "gom_ffn_fused_ln_f32": (I, [P, I, P, P, P, P, P, F, P, I, I, I, I, P, P]),
    "gom_ffn_fused_image_acc_order": (I, [P, L, I, P, P, P, L, I, I, I, P, L, P]),
    "gom_dec_tail_image_bytes": (L, [I, I, I]),
    "gom_dec_tail_f32": (I, [P, I, P, I, P, P, P, P, F, P, P, P, P, P, P, P, P, P, I, P, P, I, I, P, P]),
    "gom_dec_tail2_wave_bytes": (L, [I, I, I, I, I]),
    "gom_dec_tail2_image_lin": (I, [P, L, I, P, L, L, I, P]),
    "gom_dec_tail2_image_mlp": (I, [P, L, I, P, L, I, I, P, L, L, I, P]),
    "gom_dec_tail2_f32": (I, [P, I, P, I, P, L, I, P, P, P, P, F, P, P, P, P, P, P, F, P, P, P, P, P, P, P, P, P, P, P, P, P, I, P, P, I, I,
                              I, P, P]),
    "gom_dec_tail_lin_image_bytes": (L, []),
    "gom_dec_tail_lin_image": (I, [P, L, I, P, L, P]),
    "gom_dec_tail_proj_f32": (I, [P, I, P, I, P, I, P, P, P, P, F, P, P, P, P, F, P, P, P, P, P, P, P, P, P, I, P, P, I, I, P, P]),
    "gom_mlp2_fused_f32": (I, [P, I, P, P, P, I, P, I, I, I, I, P, P]),
    "gom_pack_records_f32": (I, [P, I, I, P, P, P, P, P, P, I, I, I, I, F, F, P, P]),
    "gom_relu_backward_f32": (I, [P, P, P, L, P]),
    "gom_softmax_rows_backward_f32": (I, [P, P, P, L, I, L, F, P]),
    "gom_asso_ce_f32": (I, [P, I, P, I, P, L, P, P, P, P]),
    "gom_sigmoid_focal_f32": (I, [P, P, F, F, L, P, P, P]),
    "gom_layernorm_f32": (I, [P, P, P, P, P, L, I, F, P]),
    "gom_groupnorm32_nhwc_f32": (I, [P, P, P, P, P, L, I, I, I, F, P]),
    "gom_mha_core_f32": (I, [P, P, P, P, I, I, I, I, I, I, ctypes.POINTER(c_long), P]),
    "gom_resample_ksize_bilinear": (I, [I, I]),
    "gom_resample_coeffs_bilinear": (I, [I, I, P, P, I]),
    "gom_resize_bilinear_u8_hwc3": (I, [P, I, I, I, P, P, I, P, P, I, P, I, I, I, P]),
    "gom_ingest_u8_hwc3_to_nhwc4": (I, [P, I, I, I, P, P, I, P, P, I, ctypes.POINTER(c_float),
                                        ctypes.POINTER(c_float), P, I, I, I, P]),
    "gom_preprocess_nchw_to_nhwc4": (I, [P, ctypes.POINTER(c_float), ctypes.POINTER(c_float), P, I, I, I, P]),
    "gom_maxpool3x3s2_nhwc_f32": (I, [P, P, I, I, I, I, P]),
    "gom_pos_encoding_2d_f32": (I, [P, P, P, I, I, P]),
    "gom_point_pos_embed_f32": (I, [P, P, P, L, P]),
    "gom_ref_sigmoid_f32": (I, [P, I, P, P, L, I, P]),
    "gom_ref_update_f32": (I, [P, I, P, P, P, P, F, F, P, P, L, P]),
    "gom_proposal_valid": (I, [P, P, I, P, L, P]),
    "gom_encoder_reference_points": (I, [P, P, I, P, L, P]),
    "gom_bezier_reference_points": (I, [P, P, P, P, I, P, P, I, L, I, I, I, P]),
    "gom_scale_xy_f32": (I, [P, L, F, F, P]),
    "gom_flag_nonfinite_f32": (I, [P, L, P, P]),
    "gom_add_f32": (I, [P, P, P, L, P]),
    "gom_copy_words": (I, [P, P, L, P]),
    "gom_broadcast_rows_f32": (I, [P, P, L, I, P]),
    "gom_topk_workspace_bytes": (L, [I, L, I]),
    "gom_topk_tokens": (I, [P, I, P, P, I, L, I, P, P, P, P]),
    "gom_argmax_rows_f32": (I, [P, I, I, L, P, P]),
    "gom_detect_post": (I, [P, I, P, I, P, P, P, I, I, I, F, F, F, F, F, P, P, P, P, P, P, P, P]),
    "gom_gather_rows_f32": (I, [P, P, P, I, I, P]),
    "gom_asso_activate_f32": (I, [P, I, P, I, I, P, I, P]),
    "gom_track_score_f32": (I, [P, I, P, P, P, F, F, I, I, I, I, F, P, P]),
    "gom_asso_score_f32": (I, [P, I, P, I, P, P, P, F, F, I, I, I, I, F, P, P]),
    "gom_short_term_pairs_f32": (I, [P, P, I, P, P, P, F, F, I, I, I, P, P]),
    "gom_mha_core_segments_f32": (I, [P, P, P, P, P, I, I, I, I, I, I, I, I, I, P]),
    "gom_pos_encoding_2d_valid_f32": (I, [P, P, P, I, I, I, I, P]),
    "gom_proposal_valid_masked": (I, [P, P, I, P, P, L, P]),
    "gom_encoder_reference_points_masked": (I, [P, P, P, I, P, L, P]),
    "gom_bezier_reference_points_masked": (I, [P, P, P, P, P, I, P, P, I, L, I, I, I, P]),
    "gom_msda_fused_forward_vr": (I, [P, I, P, P, L, I, P, P, P, P, I, I, P]),
    "gom_zero_padded_tokens_f32": (I, [P, I, I, I, P, P, P, I, I, L, P]),
    "gom_layernorm_any_f32": (I, [P, P, P, P, L, I, F, P]),
    "gom_gelu_f32": (I, [P, L, P]),
    "gom_swin_patchify_f32": (I, [P, P, I, I, I, P]),
    "gom_swin_window_gather_f32": (I, [P, P, I, I, I, I, I, P]),
    "gom_swin_window_scatter_add_f32": (I, [P, P, P, I, I, I, I, I, P]),
    "gom_swin_patch_merge_f32": (I, [P, P, I, I, I, I, P]),
    "gom_swin_window_attention_f32": (I, [P, P, P, P, L, I, I, I, P]),
    "gom_im2col_nhwc_f32": (I, [P, P, I, I, I, I, I, I, I, I, I, I, P]),
    "gom_grouped_conv3x3_nhwc_f32": (I, [P, P, P, P, P, I, P, I, I, I, I, I, I, I, P]),
    "gom_silu_f32": (I, [P, L, P]),
    "gom_vitae_window_gather_f32": (I, [P, P, I, I, I, I, P]),
    "gom_vitae_window_crop_f32": (I, [P, P, P, P, I, I, I, I, P]),
    "gom_vitae_window_attention_f32": (I, [P, P, L, I, I, P]),
    "gom_softmax_rows_scaled_f32": (I, [P, L, I, L, F, P]),
    "gom_transpose_f32": (I, [P, P, I, I, L, L, P]),
    "gom_flash_attention_f32": (I, [P, P, P, P, I, I, I, I, I, I, P, P]),
    "gom_match_workspace_floats": (L, [I, I, I, I]),
    "gom_gather_match_f32": (I, [P, I, P, I, P, I, I, I, I, P, P, P, P]),
    "gom_match_scores_proj_f32": (I, [P, I, P, I, P, P, P, P, P, I, I, I, I, I, P, I, P, I, I, I, I, F, F, I, F, P, L, P, P]),
    "gom_tracker_set_projections": (I, [P, P, I]),
    "gom_match_scores_f32": (I, [P, I, P, P, P, P, P, I, I, I, I, I, P, I, P, I, I, I, I, F, F, I, F, P, L, P, P]),
    "gom_tracker_create": (P, [I, F, I, I, I, F, P, I, P, I, I, I, I]),
    "gom_tracker_destroy": (None, [P]),
    "gom_stream_create_cu_mask": (I, [P, I, P]),
    "gom_stream_destroy": (I, [P]),
    "gom_tracker_run": (I, [P, I, P, P, P, P, I, L, P, P, P, I, F, F, P, P, P, P]),
    "gom_tracker_run_wh": (I, [P, I, P, P, P, P, I, L, P, P, P, I, P, P, P, P, P]),
    "gom_linear_sum_assignment": (I, [ctypes.POINTER(ctypes.c_double), L, L, ctypes.POINTER(c_long),
                                      ctypes.POINTER(c_long)]),
}

_lib = None


class GomError(RuntimeError):
    pass


def load():
    """Load the library once; raise loudly when it is absent (no CPU fallback exists)."""
    global _lib
    if _lib is not None:
        return _lib
    # torch first: it ships its own libamdhip64; loading ours before it would put two HIP runtimes in the process
    import torch  # noqa: F401
    if not os.path.exists(LIB_PATH):
        raise GomError("libgomatching_hip.so not built: run `python -c 'import __graft_entry__ as g; g.build()'` "
                       "(expected at %s)" % LIB_PATH)
    lib = ctypes.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)          # AttributeError if the symbol is not exported
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


def check(rc, what=""):
    if rc != GOM_OK:
        msg = _ERR.get(rc, "HIP error %d" % (rc - 1000) if rc >= 1000 else "error %d" % rc)
        raise GomError("%s failed: %s" % (what or "libgomatching_hip call", msg))
