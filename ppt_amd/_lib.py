"""ctypes binding of libppt_hip.so (include/ppt_hip.h).  The product path has NO fallback: if
the library is missing this raises, loudly, instead of computing anything on the CPU."""
import ctypes
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("PPT_HIP_LIB") or os.path.join(_HERE, "csrc", "libppt_hip.so")      # (PPT_HIP_LIB: a variant build, A/B runs)
_lib = None

PPT_F32, PPT_BF16, PPT_F16 = 0, 1, 2
A_PLAIN, A_AFFINE_RELU, A_CONV1 = 0, 1, 2
ACT_NONE, ACT_RELU, ACT_GELU, ACT_QUICKGELU = 0, 1, 2, 3

c_void_p, c_int, c_int64, c_float, c_size_t = (ctypes.c_void_p, ctypes.c_int, ctypes.c_int64,
                                               ctypes.c_float, ctypes.c_size_t)


class GemmParams(ctypes.Structure):
    """struct ppt_gemm_params (include/ppt_hip.h) -- field order must match the header."""
    _fields_ = [
        ("A", c_void_p), ("lda", c_int64), ("B", c_void_p), ("ldb", c_int64),
        ("C", c_void_p), ("ldc", c_int64), ("M", c_int), ("N", c_int), ("K", c_int),
        ("dtype", c_int), ("c_dtype", c_int),
        ("a_mode", c_int), ("a_scale", c_void_p), ("a_shift", c_void_p), ("pts", c_void_p),
        ("w1", c_void_p), ("b1", c_void_p),
        ("bias", c_void_p), ("group_add", c_void_p), ("group_rows", c_int), ("act", c_int),
        ("dact_pre", c_void_p), ("ld_dact", c_int64),
        ("row_scale", c_void_p), ("row_scale_rows", c_int),
        ("residual", c_void_p), ("ld_res", c_int64), ("residual2", c_void_p), ("ld_res2", c_int64),
        ("C2", c_void_p), ("ldc2", c_int64), ("c2_dtype", c_int), ("c2_pre", c_int),
        ("col_sum", c_void_p), ("col_sqsum", c_void_p), ("pool_max", c_void_p), ("pool_dtype", c_int), ("pool_rows", c_int), ("pool_min", c_void_p),
        ("batch", c_int), ("strideA", c_int64), ("strideB", c_int64), ("strideC", c_int64), ("wave_prio", c_int),
        ("split16", c_int), ("split_a_pow2", c_int), ("split_b_pow2", c_int), ("split_overflow", c_void_p),
    ]


class LnLinParams(ctypes.Structure):
    """struct ppt_lnlin_params (include/ppt_hip.h) -- field order must match the header."""
    _fields_ = [
        ("x", c_void_p), ("W", c_void_p), ("C", c_void_p), ("ln_w", c_void_p), ("ln_b", c_void_p), ("ln_eps", ctypes.c_float),
        ("bias", c_void_p), ("M", c_int), ("N", c_int), ("K", c_int), ("dtype", c_int), ("slices", c_int),
    ]


class TextLinParams(ctypes.Structure):
    """struct ppt_text_lin_params (include/ppt_hip.h) -- field order must match the header."""
    _fields_ = [
        ("A", c_void_p), ("lda", c_int64), ("W", c_void_p), ("bias", c_void_p), ("residual", c_void_p), ("ld_res", c_int64),
        ("C", c_void_p), ("ldc", c_int64), ("M", c_int), ("N", c_int), ("K", c_int),
        ("split_a_pow2", c_int), ("split_b_pow2", c_int), ("split_overflow", c_void_p), ("wave_prio", c_int),
    ]


class TextMlpParams(ctypes.Structure):
    """struct ppt_text_mlp_params (include/ppt_hip.h) -- field order must match the header."""
    _fields_ = [
        ("A", c_void_p), ("lda", c_int64), ("W1", c_void_p), ("W2", c_void_p), ("b1", c_void_p), ("pre", c_void_p), ("parts", c_void_p),
        ("M", c_int), ("D", c_int), ("hidden", c_int), ("mode", c_int), ("dtype", c_int), ("wave_prio", c_int),
        ("ln_w", c_void_p), ("ln_b", c_void_p), ("ln_eps", ctypes.c_float), ("ln_mean", c_void_p), ("ln_rstd", c_void_p),
        ("split_a_pow2", c_int), ("split_b_pow2", c_int), ("split_overflow", c_void_p),
    ]


class VitMlpParams(ctypes.Structure):
    """struct ppt_vit_mlp_params (include/ppt_hip.h) -- field order must match the header."""
    _fields_ = [
        ("x", c_void_p), ("out", c_void_p), ("W1", c_void_p), ("W2", c_void_p), ("ln_w", c_void_p), ("ln_b", c_void_p),
        ("ln_eps", c_float), ("b1", c_void_p), ("b2", c_void_p), ("row_scale", c_void_p), ("row_scale_rows", c_int),
        ("residual2", c_void_p), ("M", c_int), ("D", c_int), ("hidden", c_int), ("workgroups", c_int), ("n_chunks", c_int),
        ("rows_per_chunk", c_int), ("dtype", c_int),
        ("proj_a", c_void_p), ("proj_W", c_void_p), ("proj_b", c_void_p), ("proj_row_scale", c_void_p), ("proj_row_scale_rows", c_int),
    ]


class BallMulti(ctypes.Structure):
    """struct ppt_ball_multi (include/ppt_hip.h)."""
    _fields_ = [("n", c_int), ("r2", c_float * 3), ("K", c_int * 3), ("idx", c_void_p * 3), ("gxyz", c_void_p * 3)]


class AdamwTensor(ctypes.Structure):
    """struct ppt_adamw_tensor (include/ppt_hip.h)."""
    _fields_ = [("p", c_void_p), ("g", c_void_p), ("exp_avg", c_void_p), ("exp_avg_sq", c_void_p), ("n", c_int64), ("step", c_int)]


class WprepItem(ctypes.Structure):
    """struct ppt_wprep_item (include/ppt_hip.h)."""
    _fields_ = [("w", c_void_p), ("ldw", c_int64), ("N", c_int), ("col0", c_int), ("K", c_int), ("sub_col0", c_int), ("Kp", c_int),
                ("out", c_void_p), ("out_t", c_void_p)]


class RowGemmParams(ctypes.Structure):
    """struct ppt_rowgemm_params (include/ppt_hip.h) -- field order must match the header."""
    _fields_ = [
        ("A", c_void_p), ("W", c_void_p), ("C", c_void_p), ("C2", c_void_p), ("M", c_int), ("N", c_int), ("K", c_int),
        ("a_ln", c_int), ("ln_w", c_void_p), ("ln_b", c_void_p), ("ln_eps", c_float), ("ln_mean", c_void_p), ("ln_rstd", c_void_p), ("bias", c_void_p), ("act", c_int),
        ("residual_form", c_int), ("residual", c_void_p), ("residual2", c_void_p), ("row_scale", c_void_p),
        ("row_scale_rows", c_int), ("walkers", c_int), ("groups", c_int), ("dtype", c_int),
    ]


_SIGNATURES = {
    "ppt_abi_version": (c_int, []),
    "ppt_cross_entropy_rows": (c_int, [c_void_p, c_void_p, c_float, c_int64, c_int, c_int64, c_void_p, c_void_p, c_void_p, c_void_p]),
    "ppt_gn_finish": (c_int, [c_void_p, c_int, c_int, c_int, ctypes.c_double, ctypes.c_double, c_int, c_void_p, c_void_p, c_void_p]),
    "ppt_set_gemm256": (None, [c_int]),
    "ppt_get_gemm256": (c_int, []),
    "ppt_set_wave_priority": (None, [c_int]),
    "ppt_get_wave_priority": (c_int, []),
    "ppt_set_persistent_occupancy": (None, [c_int]),
    "ppt_get_persistent_occupancy": (c_int, []),
    "ppt_rows_matmul_f32": (c_int, [c_void_p, c_void_p, c_int, c_int, c_int, c_float, c_void_p, c_void_p]),
    "ppt_fps_f32": (c_int, [c_void_p, c_int, c_int, c_int, c_void_p, c_void_p, c_void_p, c_void_p]),
    "ppt_knn_group_f32": (c_int, [c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_void_p, c_void_p, c_void_p, c_void_p]),
    "ppt_square_distance_f32": (c_int, [c_void_p, c_void_p, c_int, c_int, c_int, c_void_p, c_void_p]),
    "ppt_ball_query_multi_f32": (c_int, [c_void_p, c_void_p, c_int, c_int, c_int, ctypes.POINTER(BallMulti), c_void_p]),
    "ppt_ball_query_f32": (c_int, [c_void_p, c_void_p, c_int, c_int, c_int, c_float, c_int, c_void_p, c_void_p, c_void_p]),
    "ppt_gather_add": (c_int, [c_void_p, c_int, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int, c_void_p, c_int,
                               c_void_p, c_void_p, c_void_p]),
    "ppt_pool_finish": (c_int, [c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_void_p, c_void_p, c_void_p, c_int, c_int64,
                                c_void_p]),
    "ppt_bn_act_rows": (c_int, [c_void_p, c_int, c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_void_p]),
    "ppt_gemm": (c_int, [ctypes.POINTER(GemmParams), c_void_p]),
    "ppt_gemm256": (c_int, [ctypes.POINTER(GemmParams), c_void_p]),
    "ppt_three_nn_interp_fwd": (c_int, [c_void_p, c_int, c_void_p, c_int, c_void_p, c_void_p, c_int, c_int, c_int, c_void_p, c_int,
                                        c_int, c_void_p, c_void_p]),
    "ppt_scatter_rows_bwd": (c_int, [c_void_p, c_void_p, c_void_p, c_int64, c_int, c_int, c_int, c_int, c_int, c_int, c_void_p,
                                     c_void_p]),
    "ppt_sum_groups": (c_int, [c_void_p, c_int64, c_int, c_int, c_void_p, c_void_p]),
    "ppt_vit_mlp_retile": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p]),
    "ppt_vit_proj_retile": (c_int, [c_void_p, c_void_p, c_void_p]),
    "ppt_vit_mlp_bf16": (c_int, [ctypes.POINTER(VitMlpParams), c_void_p]),
    "ppt_lnlin_retile": (c_int, [c_void_p, c_void_p, c_int, c_void_p]),
    "ppt_lnlin": (c_int, [ctypes.POINTER(LnLinParams), c_void_p]),
    "ppt_text_mlp_retile": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p]),
    "ppt_text_mlp_pair": (c_int, [ctypes.POINTER(TextMlpParams), c_void_p]),
    "ppt_text_mlp_retile_split": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_void_p]),
    "ppt_text_lin_retile_split": (c_int, [c_void_p, c_void_p, c_int, c_int, c_int, c_void_p]),
    "ppt_text_lin_split": (c_int, [ctypes.POINTER(TextLinParams), c_void_p]),
    "ppt_vit_mlp3_retile": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p]),
    "ppt_vit_mlp3_bf16": (c_int, [ctypes.POINTER(VitMlpParams), c_void_p]),
    "ppt_rowgemm_bf16": (c_int, [ctypes.POINTER(RowGemmParams), c_void_p]),
    "ppt_layernorm_fwd": (c_int, [c_void_p, c_void_p, c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_int,
                                  c_void_p, c_void_p, c_int, c_int, c_float, c_void_p]),
    "ppt_layernorm_bwd": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int,
                                  c_void_p, c_int, c_void_p, c_void_p, c_int, c_int, c_int, c_void_p]),
    "ppt_layernorm_fwd_sum": (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_int,
                                      c_void_p, c_void_p, c_int, c_int, c_float, c_void_p]),
    "ppt_layernorm_bwd_sum": (c_int, [c_void_p, c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int,
                                      c_void_p, c_int, c_int, c_int, c_void_p]),
    "ppt_col_sums": (c_int, [c_void_p, c_int, c_int, c_int, c_int64, c_void_p, c_void_p]),
    "ppt_attention_fwd": (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_float, c_int,
                                  c_int, c_void_p]),
    "ppt_attention_bwd": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int,
                                  c_int, c_int, c_float, c_int, c_int, c_void_p]),
    "ppt_attention_prefix_workspace_bytes": (ctypes.c_size_t, [c_int, c_int, c_int, c_int]),
    "ppt_pointmlp_cloud_rstd": (c_int, [c_void_p, c_int, c_int, ctypes.c_double, c_void_p, c_void_p]),
    "ppt_pointmlp_pq": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_void_p]),
    "ppt_attention_bwd_split16": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int, c_float, c_int, c_void_p]),
    "ppt_attention_fwd_split16": (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int, c_float, c_int, c_void_p]),
    "ppt_attention_prefix_fwd": (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int, c_float, c_int, c_void_p]),
    "ppt_attention_prefix_bwd": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int,
                                         c_int, c_int, c_float, c_int, c_void_p]),
    "ppt_conv1_stats": (c_int, [c_void_p, c_int64, c_void_p, c_void_p, c_int, c_void_p, c_void_p,
                                ctypes.POINTER(c_int), c_void_p]),
    "ppt_conv1_stats_max_partials": (c_int, [c_int64]),
    "ppt_conv1_stats_rows_per_partial": (c_int, []),
    "ppt_bn_finalize": (c_int, [c_void_p, c_void_p, c_int, c_int, c_int64, c_int, c_void_p, c_void_p, c_float, c_int,
                                c_float, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p]),
    "ppt_bn_finalize_ws": (c_int, [c_void_p, c_void_p, c_int, c_int, c_int64, c_int, c_void_p, c_void_p, c_float, c_int,
                                   c_float, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p,
                                   ctypes.c_size_t, c_void_p]),
    "ppt_rows_stats_rows_per_partial": (c_int, []),
    "ppt_rows_stats_f32": (c_int, [c_void_p, c_int64, c_int, c_void_p, c_void_p, c_void_p]),
    "ppt_bn_rows_bwd_reduce": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int64, c_int,
                                       c_void_p, c_void_p, c_void_p]),
    "ppt_bn_rows_bwd_apply": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int,
                                      c_int, c_int64, c_int, c_void_p, c_void_p, c_int, c_void_p]),
    "ppt_bn_finalize_workspace_bytes": (ctypes.c_size_t, [c_int, c_int]),
    "ppt_gn_stats_chunks": (c_int, [c_int]),
    "ppt_gn_bwd_chunks": (c_int, [c_int]),
    "ppt_gn_stats": (c_int, [c_void_p, c_int, c_int, c_int, c_int, c_void_p, c_void_p]),
    "ppt_gn_lrelu_max": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int, c_float,
                                 c_void_p, c_void_p, c_void_p]),
    "ppt_gn_bwd_sums": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int,
                                c_int, c_float, c_void_p, c_void_p, c_void_p]),
    "ppt_gn_bwd_apply": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int,
                                 c_int, c_int, c_int, c_float, c_void_p, c_void_p]),
    "ppt_mini_pointnet_conv12_bf16": (c_int, [c_void_p, c_int64, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_void_p, c_void_p,
                                              c_int, c_void_p, c_void_p, c_void_p]),
    "ppt_mini_pointnet_conv3_bf16": (c_int, [c_void_p, c_int64, c_int, c_void_p, c_void_p, c_int, c_void_p, c_void_p, c_void_p,
                                             c_void_p]),
    "ppt_mini_pointnet_conv4_bf16": (c_int, [c_void_p, c_int64, c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_void_p,
                                             c_void_p]),
    "ppt_mini_pointnet_conv12_half": (c_int, [c_void_p, c_int64, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_void_p, c_void_p,
                                              c_int, c_void_p, c_void_p, c_int, c_void_p]),
    "ppt_mini_pointnet_conv3_half": (c_int, [c_void_p, c_int64, c_int, c_void_p, c_void_p, c_int, c_void_p, c_void_p, c_void_p,
                                             c_int, c_void_p]),
    "ppt_mpn34_retile": (c_int, [c_void_p, c_void_p, c_void_p]),
    "ppt_mini_pointnet_conv34_half": (c_int, [c_void_p, c_int64, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_void_p]),
    "ppt_mini_pointnet_conv4_half": (c_int, [c_void_p, c_int64, c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_void_p,
                                             c_int, c_void_p]),
    "ppt_conv12_stats_bf16": (c_int, [c_void_p, c_int64, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_void_p, c_void_p,
                                      c_int, c_void_p, c_void_p, c_void_p, c_void_p]),
    "ppt_affine_conv_pool_bf16": (c_int, [c_void_p, c_int64, c_int64, c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int,
                                          c_void_p, c_void_p, c_void_p, c_void_p, c_void_p]),
    "ppt_group_anchor_stats": (c_int, [c_void_p, c_int, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int, c_void_p, c_void_p]),
    "ppt_bn_res_act_rows": (c_int, [c_void_p, c_int, c_void_p, c_int, c_int64, c_int, c_int, c_void_p, c_void_p, c_void_p, c_void_p,
                                    c_void_p, c_int, c_void_p]),
    "ppt_gemm_tn_bf16": (c_int, [c_void_p, c_int64, c_void_p, c_int64, c_int64, c_int, c_int, c_int, c_void_p, c_void_p]),
    "ppt_gemm_tn_half": (c_int, [c_void_p, c_int64, c_void_p, c_int64, c_int64, c_int, c_int, c_int, c_void_p, c_int, c_void_p]),
    "ppt_head_logits": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_void_p, c_void_p, c_void_p]),
    "ppt_head_ce_bwd": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_float, c_int, c_int, c_int, c_void_p, c_void_p,
                                c_void_p]),
    "ppt_linear3_gelu": (c_int, [c_void_p, c_int64, c_void_p, c_void_p, c_int, c_void_p, c_int, c_void_p]),
    "ppt_cls_max_pool": (c_int, [c_void_p, c_int, c_int, c_int, c_int, c_void_p, c_void_p, c_void_p]),
    "ppt_adamw_step": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int64, c_float, c_float, c_float, c_float, c_float, c_int,
                               c_float, c_void_p, c_void_p]),
    "ppt_adamw_multi": (c_int, [c_void_p, c_int, c_float, c_float, c_float, c_float, c_float, c_float, c_void_p, c_void_p]),
    "ppt_prompt_rows": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_void_p, c_void_p]),
    "ppt_prompt_rows_bwd": (c_int, [c_void_p, c_void_p, c_int, c_int, c_int, c_float, c_void_p, c_void_p]),
    "ppt_convert": (c_int, [c_void_p, c_int, c_void_p, c_int, c_int64, c_void_p]),
    "ppt_weights_prep": (c_int, [c_void_p, c_int, c_int, c_void_p]),
    "ppt_health_check": (c_int, [c_void_p, c_int, c_int64, c_void_p, ctypes.c_uint32, c_void_p, c_void_p]),
    "ppt_labels_check": (c_int, [c_void_p, c_int64, c_int64, c_int64, c_void_p, ctypes.c_uint32, c_void_p]),
    "ppt_scale_rows_convert": (c_int, [c_void_p, c_int64, c_int, c_int, c_void_p, c_void_p, c_int, c_void_p, c_void_p, c_void_p, c_void_p]),
    "ppt_convert_scaled": (c_int, [c_void_p, c_int, c_void_p, c_int, c_int64, c_float, c_void_p]),
    "ppt_transpose": (c_int, [c_void_p, c_int, c_void_p, c_int, c_int, c_int, c_int64, c_void_p]),
    "ppt_reduce_rows": (c_int, [c_void_p, c_int, c_int, c_void_p, c_int, c_void_p]),
}


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(
                f"{LIB_PATH} not found: the HIP extension is not built (run `python -m ppt_amd.build` "
                "or __graft_entry__.build()).  ppt_amd has no CPU fallback.")
        L = ctypes.CDLL(LIB_PATH)
        L.ppt_strerror.restype = ctypes.c_char_p
        L.ppt_strerror.argtypes = [c_int]
        for name, (res, args) in _SIGNATURES.items():
            fn = getattr(L, name)
            fn.restype, fn.argtypes = res, args
        _lib = L
    return _lib


def check(code, what=""):
    if code != 0:
        raise RuntimeError(f"libppt_hip: {what} failed: {lib().ppt_strerror(code).decode()} ({code})")
