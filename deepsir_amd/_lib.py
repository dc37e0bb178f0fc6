"""ctypes binding of libdsir.so (the C ABI declared in include/dsir.h).

There is no CPU fallback: if the shared library is missing or does not load,
importing the engine fails loudly.  Build it with ``python -c "import
__graft_entry__ as g; g.build()"`` or ``python deepsir_amd/csrc/build.py``.
"""
from __future__ import annotations

import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# DSIR_LIB selects another build of the SAME library (kernel-variant A/B runs); never a different backend.
LIB_PATH = os.environ.get("DSIR_LIB", os.path.join(_HERE, "libdsir.so"))

c_float_p = C.POINTER(C.c_float)
c_i32_p = C.POINTER(C.c_int32)
c_i64_p = C.POINTER(C.c_int64)


class dsir_cfg(C.Structure):
    _fields_ = [
        ("feat_len", C.c_int32),
        ("num_knn", C.c_int32),
        ("num_layers", C.c_int32),
        ("sub_sampling_ratio", C.c_int32 * 4),
        ("d_out", C.c_int32 * 4),
        ("out_feat_dim", C.c_int32),
        ("num_classes", C.c_int32),
        ("max_points", C.c_int32),
        ("max_pairs", C.c_int32),
        ("pipeline", C.c_int32),
    ]


class dsir_pair_batch(C.Structure):
    _fields_ = [
        ("pairs", C.c_int32), ("n_src", C.c_int32), ("n_ref", C.c_int32),
        ("points_src", C.c_void_p), ("points_ref", C.c_void_p),
        ("src_xyz", C.c_void_p), ("src_neigh", C.c_void_p), ("src_sub", C.c_void_p), ("src_interp", C.c_void_p),
        ("ref_xyz", C.c_void_p), ("ref_neigh", C.c_void_p), ("ref_sub", C.c_void_p), ("ref_interp", C.c_void_p),
        ("forced_idx", C.c_void_p),
    ]


class dsir_pair_result(C.Structure):
    _fields_ = [
        ("transforms", C.c_void_p), ("idx", C.c_void_p), ("logits", C.c_void_p), ("pt_ref_new", C.c_void_p),
        ("invalid", C.c_void_p), ("desc_src", C.c_void_p), ("desc_ref", C.c_void_p),
    ]


class dsir_cloud_out(C.Structure):
    _fields_ = [
        ("xyz", C.c_void_p), ("feat", C.c_void_p), ("logits", C.c_void_p), ("score", C.c_void_p), ("label", C.c_void_p),
        ("index", C.c_void_p),
    ]


# every symbol include/dsir.h declares: (restype, argtypes)
SYMBOLS = {
    "dsir_create": (C.c_int, [C.c_int, C.POINTER(dsir_cfg), C.POINTER(C.c_void_p)]),
    "dsir_destroy": (None, [C.c_void_p]),
    "dsir_last_error": (C.c_char_p, [C.c_void_p]),
    "dsir_stream": (C.c_void_p, [C.c_void_p]),
    "dsir_sync": (C.c_int, [C.c_void_p]),
    "dsir_set_stream": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int]),
    "dsir_num_weights": (C.c_int, [C.c_void_p]),
    "dsir_weight_name": (C.c_char_p, [C.c_void_p, C.c_int, c_i64_p]),
    "dsir_load_weight": (C.c_int, [C.c_void_p, C.c_char_p, C.c_void_p, c_i64_p, C.c_int]),
    "dsir_finalize_weights": (C.c_int, [C.c_void_p]),
    "dsir_narrow_i64": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64]),
    "dsir_knn_pyramid": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p,
                                   C.c_void_p, C.c_void_p]),
    "dsir_randla_forward": (C.c_int, [C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p,
                                      C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "dsir_score": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p, C.c_int64,
                             C.c_int, C.c_int, C.c_void_p, C.c_void_p]),
    "dsir_aggregate": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p, C.c_int, C.c_int,
                                 C.c_void_p]),
    "dsir_nn_match": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p]),
    "dsir_nn_match_screened": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p, c_i64_p]),
    "dsir_kabsch": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p,
                              C.c_void_p]),
    "dsir_register": (C.c_int, [C.c_void_p, C.POINTER(dsir_pair_batch), C.c_int, C.POINTER(dsir_pair_result)]),
    "dsir_forward_pair": (C.c_int, [C.c_void_p, C.POINTER(dsir_pair_batch), C.c_int, C.POINTER(dsir_cloud_out),
                                    C.POINTER(dsir_cloud_out)]),
    "dsir_icp_refine": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_float, C.c_int,
                                  C.c_float, C.c_float, C.c_void_p, C.c_void_p, C.c_void_p]),
    "dsir_pose_finetune": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_float,
                                     C.c_int, C.c_float, C.c_int, C.c_void_p, C.c_void_p]),
    "dsir_align_loss_backward": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int,
                                           C.c_int, C.c_int, C.c_int, C.c_int, C.c_float, C.c_float, C.c_float, C.c_void_p,
                                           C.POINTER(C.c_double), C.c_void_p]),
    "dsir_align_loss_backward2": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int,
                                            C.c_int, C.c_int, C.c_int, C.c_int, C.c_float, C.c_float, C.c_float, C.c_void_p,
                                            C.POINTER(C.c_double), C.c_void_p, C.POINTER(C.c_double)]),
    "dsir_match_timer": (C.c_int, [C.c_void_p, C.c_int, C.POINTER(C.c_double), c_i64_p]),
    "dsir_match_timer2": (C.c_int, [C.c_void_p, C.c_int, C.POINTER(C.c_double), C.POINTER(C.c_double), c_i64_p]),
    "dsir_enable_screen": (C.c_int, [C.c_void_p, C.c_int]),
    "dsir_enable_agg_split": (C.c_int, [C.c_void_p, C.c_int]),
    "dsir_split_f16": (None, [C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p]),
    "dsir_match_timer_device": (C.c_int, [C.c_void_p, C.c_int, C.POINTER(C.c_double), c_i64_p]),
    "dsir_enable_match_timer": (C.c_int, [C.c_void_p, C.c_int]),
    "dsir_enable_graph": (C.c_int, [C.c_void_p, C.c_int]),
    "dsir_graph_stats": (C.c_int, [C.c_void_p, c_i64_p]),
    "dsir_enable_walk": (C.c_int, [C.c_void_p, C.c_int]),
    "dsir_enable_fork": (C.c_int, [C.c_void_p, C.c_int]),
    "dsir_walk_trace": (C.c_int, [C.c_void_p, C.c_int, c_i64_p, c_i64_p]),
    "dsir_screen_stats": (C.c_int, [C.c_void_p, C.c_int, c_i64_p]),
    "dsir_prune_stats": (C.c_int, [C.c_void_p, C.c_int, c_i64_p]),
    "dsir_set_prune_thresholds": (C.c_int, [C.c_void_p, C.c_int, C.c_int64]),
    "dsir_set_kabsch_chunked_min": (C.c_int, [C.c_void_p, C.c_int]),
    "dsir_set_tuning": (None, [C.c_int]),
    "dsir_tuning": (C.c_int, []),
    "dsir_gn_contributions": (C.c_int, [C.POINTER(dsir_cfg), C.c_int]),
    "dsir_gn_contribution_limit": (C.c_int, []),
    "dsir_max_points_limit": (C.c_int, [C.POINTER(dsir_cfg)]),
    "dsir_screen_bounds": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p,
                                     C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "dsir_screen_cap": (C.c_int, []),
    "dsir_voxel_downsample": (C.c_int, [C.c_void_p, C.c_void_p, c_i64_p, C.c_int, C.c_int, C.c_float, c_float_p, C.c_int,
                                        C.c_void_p, C.c_void_p]),
    "dsir_resample": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_uint64,
                                C.c_void_p]),
    "dsir_eval_metrics": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int,
                                    C.c_int, C.c_int, C.c_float, C.c_float, C.c_void_p]),
    # include/dsir_train.h: the training operators (first argument: hipStream_t)
    "dsir_t_gemm": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_int,
                              C.c_int64, C.c_int, C.c_int, C.c_float]),
    "dsir_t_gemm_dw_scratch": (C.c_size_t, [C.c_int64, C.c_int, C.c_int]),
    "dsir_t_gemm_dw": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_int64, C.c_int, C.c_int, C.c_void_p,
                                 C.c_void_p, C.c_void_p]),
    "dsir_t_gn_scratch": (C.c_size_t, [C.c_int, C.c_int, C.c_int]),
    "dsir_t_gn_fwd": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_int,
                                C.c_void_p, C.c_void_p, C.c_void_p]),
    "dsir_t_gn_bwd": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p,
                                C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "dsir_t_bn_running": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int64, C.c_float, C.c_void_p, C.c_void_p]),
    "dsir_t_gather": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_int, C.c_int]),
    "dsir_t_scatter_plan_scratch": (C.c_size_t, [C.c_int64]),
    "dsir_t_scatter_plan": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]),
    "dsir_t_scatter_add": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_int,
                                     C.c_int]),
    "dsir_t_relpos": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p]),
    "dsir_t_inlier_input": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int,
                                      C.c_void_p]),
    "dsir_t_attpool_fwd": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_int, C.c_int, C.c_void_p]),
    "dsir_t_attpool_bwd": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_int, C.c_int, C.c_void_p,
                                     C.c_void_p]),
    "dsir_t_maxpool_fwd": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p,
                                     C.c_void_p]),
    "dsir_t_maxpool_bwd": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int,
                                     C.c_void_p, C.c_int]),
    "dsir_t_weighted_ce_scratch": (C.c_size_t, [C.c_int64]),
    "dsir_t_weighted_ce": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_int, C.c_float, C.c_void_p,
                                     C.c_void_p, C.c_void_p]),
    "dsir_t_l2norm_fwd": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int64, C.c_int, C.c_void_p, C.c_void_p]),
    "dsir_t_l2norm_bwd": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_int, C.c_void_p]),
    "dsir_t_det_des_loss_scratch": (C.c_size_t, [C.c_int, C.c_int]),
    "dsir_t_det_des_loss": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int,
                                      C.c_int, C.c_float, C.c_float, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "dsir_t_add_leaky_fwd": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p]),
    "dsir_t_add_leaky_bwd": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p]),
    "dsir_t_mul_mask": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_float, C.c_int64, C.c_void_p]),
    "dsir_t_topk_scratch": (C.c_size_t, [C.c_int, C.c_int]),
    "dsir_t_topk": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]),
    "dsir_t_sigmoid": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p]),
    "dsir_t_axpy": (C.c_int, [C.c_void_p, C.c_float, C.c_void_p, C.c_int64, C.c_void_p]),
    "dsir_t_any_nan": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p]),
    "dsir_t_adam": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_float, C.c_float, C.c_float,
                              C.c_float, C.c_int]),
}

_lib = None


def load() -> C.CDLL:
    """Load libdsir.so and bind every declared entry point; raises if anything is missing."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            f"{LIB_PATH} not found: the HIP engine is not built.  Run `python deepsir_amd/csrc/build.py` "
            "(needs hipcc; cross-compiles gfx950 without a GPU).  There is no CPU fallback.")
    lib = C.CDLL(LIB_PATH)
    for name, (res, args) in SYMBOLS.items():
        fn = getattr(lib, name)  # AttributeError if the symbol is not exported
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib
