"""ctypes binding of libtgnn.so (the C ABI declared in include/tgnn.h).

The library is the product: there is NO CPU or eager-PyTorch fallback anywhere in this package.
If the shared object has not been built, importing this module raises with the build command.
"""
from __future__ import annotations

import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("TGNN_LIB_PATH") or os.path.join(_HERE, "libtgnn.so")   # override: kernel experiments only

ACT_NONE, ACT_LEAKY_RELU, ACT_SIGMOID = 0, 1, 2
BN_MAX_PARTIALS = 512
BN_STAT_ROWS = 4


class TgnnError(RuntimeError):
    code = 0          # the library's status (include/tgnn.h: TGNN_ERR_*)


class ModelDims(C.Structure):
    """tgnn_model_dims"""
    _fields_ = [("node_features_dim", C.c_int32), ("adj_edge_features_dim", C.c_int32),
                ("network_width", C.c_int32), ("network_depth", C.c_int32), ("output_dim", C.c_int32)]


class Graph(C.Structure):
    """tgnn_graph"""
    _fields_ = [("n_nodes", C.c_int64), ("n_adj_edges", C.c_int64), ("n_col_edges", C.c_int64),
                ("n_types", C.c_int32),
                ("adj_rowptr", C.c_void_p), ("adj_src", C.c_void_p), ("adj_type", C.c_void_p),
                ("type_rep_edge", C.c_void_p), ("col_rowptr", C.c_void_p), ("col_src", C.c_void_p),
                ("nn_tile_col_ptr", C.c_void_p), ("nn_col_meta", C.c_void_p), ("nn_col_src", C.c_void_p),
                ("nn_max_in_degree", C.c_int32),
                ("nn_mid_tile_nb", C.c_void_p), ("nn_mid_ent", C.c_void_p),
                ("nn_tile_grp_ptr", C.c_void_p), ("nn_grp", C.c_void_p), ("nn_mid_verdict", C.c_void_p)]


class TrainSave(C.Structure):
    """tgnn_train_save"""
    _fields_ = [("init_a", C.c_void_p * 2), ("init_stat", C.c_void_p * 2), ("a1", C.c_void_p), ("a2", C.c_void_p),
                ("u", C.c_void_p), ("stat1", C.c_void_p), ("stat2", C.c_void_p), ("fin_a", C.c_void_p * 4),
                ("fin_stat", C.c_void_p * 4), ("skip", C.c_void_p), ("wtab", C.c_void_p)]


class TrainGraphDesc(C.Structure):
    """tgnn_train_graph"""
    _fields_ = [("adjT_rowptr", C.c_void_p), ("adjT_src", C.c_void_p), ("adjT_type", C.c_void_p),
                ("colT_rowptr", C.c_void_p), ("colT_src", C.c_void_p), ("deg", C.c_void_p), ("inv_deg", C.c_void_p)]


ALLREDUCE_CB = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p)
ALLTOALL_CB = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_void_p)


class ShardDesc(C.Structure):
    """tgnn_shard"""
    _fields_ = [("n_own", C.c_int64), ("n_rows", C.c_int64), ("n_total", C.c_int64),
                ("send_idx", C.c_void_p), ("n_send", C.c_int64),
                ("sum_buf", C.c_void_p), ("send_buf", C.c_void_p), ("recv_buf", C.c_void_p),
                ("allreduce_f64", ALLREDUCE_CB), ("alltoall_rows", ALLTOALL_CB), ("ctx", C.c_void_p),
                ("world", C.c_int32), ("rank", C.c_int32), ("send_idx_fused", C.c_void_p), ("recv_idx_fused", C.c_void_p),
                ("side_stream", C.c_void_p), ("rccl_comm", C.c_void_p), ("rccl_comm_side", C.c_void_p),
                ("send_counts", C.c_void_p), ("recv_counts", C.c_void_p),
                ("send_row_ptr", C.c_void_p), ("send_row_slot", C.c_void_p)]


def _load() -> C.CDLL:
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            f"{LIB_PATH} is missing: the HIP extension has not been built. Build it with\n"
            f"    python -c 'import __graft_entry__ as g; g.build()'    (or: make -C tilingnn_amd/csrc)\n"
            "tilingnn_amd has no CPU fallback by design.")
    # torch first: its wheel carries its own HIP runtime, and a process must have ONE.  Loaded the other way round, the
    # library binds /opt/rocm's libamdhip64, torch then brings its copy, and whichever initialises second reports "no
    # ROCm-capable device" (seen with build() followed by smoke() in one process).
    import torch  # noqa: F401
    lib = C.CDLL(LIB_PATH)
    p, i32, i64, f32, sz = C.c_void_p, C.c_int32, C.c_int64, C.c_float, C.c_size_t
    pi32 = C.POINTER(C.c_int32)
    sigs = {
        "tgnn_version": (C.c_int, []),
        "tgnn_last_error": (C.c_char_p, []),
        "tgnn_csr_workspace_bytes": (sz, [i64, i64]),
        "tgnn_csr_build": (C.c_int, [p, i64, i64, i64, C.c_int, p, p, p, p, p, sz, p]),
        "tgnn_edge_dedup_workspace_bytes": (sz, [i64, i32]),
        "tgnn_edge_type_dedup": (C.c_int, [p, i64, i32, p, p, p, p, sz, p]),
        "tgnn_gather_i32": (C.c_int, [p, i64, p, i64, p, p]),
        "tgnn_edge_weight_table": (C.c_int, [p, p, i32, i32, p, p, p, p, p, p, i32, p, p]),
        "tgnn_nnconv_mean_fwd": (C.c_int, [p, i64, p, p, p, p, i32, p, p, i64, i32, i32, p, p, pi32, p]),
        "tgnn_nnconv_cols_max_columns": (i64, [i64, i64]),
        "tgnn_nnconv_cols_workspace_bytes": (sz, [i64]),
        "tgnn_nnconv_cols_build": (C.c_int, [p, p, p, i64, i32, p, p, p, p, sz, p]),
        "tgnn_nnconv_cols_max_types": (i32, []),
        "tgnn_nnconv_weight_image_floats": (sz, [i32]),
        "tgnn_nnconv_mean_cols_fwd": (C.c_int, [p, i64, p, p, p, p, i32, p, p, i64, i32, i32, p, p, p, pi32, p]),
        "tgnn_nnconv_mean_cols_f16_fwd": (C.c_int, [p, i64, i64, p, p, p, p, i32, p, p, i64, i32, i32, p, p, p, p, pi32, p]),
        "tgnn_nnconv_eg_max_groups": (i64, [i64, i64, i32]),
        "tgnn_nnconv_eg_build": (C.c_int, [p, p, p, i64, i32, p, p, p, sz, p]),
        "tgnn_nnconv_mean_eg_fwd": (C.c_int, [p, i64, i64, p, p, p, i32, p, p, i64, i32, p, p, p, p, pi32, p]),
        "tgnn_ubench_row_gather": (C.c_int, [i32, p, i64, p, i32, i32, C.POINTER(C.c_double), p]),
        "tgnn_mid_entries_words": (i64, [i64]),
        "tgnn_mid_entries_build": (C.c_int, [p, p, p, i64, p, p, p, p, p]),
        "tgnn_forward_path_counts": (None, [p]),
        "tgnn_set_mid_layout_limit": (None, [i64]),
        "tgnn_get_mid_layout_limit": (i64, []),
        "tgnn_mid_layout_max_nodes": (i64, []),
        "tgnn_spin_error_poll": (C.c_int, [p, C.POINTER(C.c_uint32)]),
        "tgnn_set_spin_budget_us": (C.c_uint64, [C.c_uint64]),
        "tgnn_persist_fallback": (None, [i64]),
        "tgnn_spin_error_peek": (C.c_uint32, []),
        "tgnn_gin_fwd": (C.c_int, [p, i64, p, p, p, p, p, p, p, p, p, p, i64, i32, i32, p, p, p, pi32, p]),
        "tgnn_dense_act_fwd": (C.c_int, [p, i64, i64, p, p, p, i64, i32, i32, i32, p, i64, p, pi32, p]),
        "tgnn_dense_act_slots_fwd": (C.c_int, [p, i32, i64, p, p, p, i64, i32, i32, i32, p, i64, p, pi32, p]),
        "tgnn_dense_act_slots_f16_fwd": (C.c_int, [p, i32, i64, p, p, i64, i32, i32, i32, p, i64, p, p, p, pi32, p]),
        "tgnn_bn_finalize": (C.c_int, [i32, p, i32, p, i32, i64, p, p, f32, f32, p, p, p, p, p]),
        "tgnn_bn_apply": (C.c_int, [p, i64, p, i64, i32, p, i64, p]),
        "tgnn_merge_fwd": (C.c_int, [p, p, p, p, p, i64, i32, p, p, p]),
        "tgnn_param_count": (i32, [C.POINTER(ModelDims)]),
        "tgnn_param_name": (C.c_int, [C.POINTER(ModelDims), i32, C.c_char_p, sz]),
        "tgnn_forward_workspace_bytes": (sz, [C.POINTER(ModelDims), i64, i32]),
        "tgnn_forward": (C.c_int, [C.POINTER(ModelDims), C.POINTER(C.c_void_p), p, p, C.POINTER(Graph), i32, i32,
                                   p, p, sz, p, p]),
        "tgnn_forward_begin": (C.c_int, [C.POINTER(ModelDims), C.POINTER(C.c_void_p), p, i64, i32, p, sz, p, p]),
        "tgnn_forward_resume": (C.c_int, [C.POINTER(ModelDims), C.POINTER(C.c_void_p), p, p, C.POINTER(Graph), i32, p, p, sz, p, p]),
        "tgnn_forward_begin_weights": (C.c_int, [C.POINTER(ModelDims), C.POINTER(C.c_void_p), p, p, p, i64, p, sz, p]),
        "tgnn_forward_small_prepass": (C.c_int, [C.POINTER(ModelDims), C.POINTER(C.c_void_p), p, p, p, i64, p, sz, p]),
        "tgnn_forward_sharded_workspace_bytes": (sz, [C.POINTER(ModelDims), i64, i64, i32]),
        "tgnn_forward_sharded": (C.c_int, [C.POINTER(ModelDims), C.POINTER(C.c_void_p), p, p, C.POINTER(Graph),
                                           C.POINTER(ShardDesc), i32, p, p, sz, p]),
        "tgnn_sublayout_workspace_bytes": (sz, [i64, i64, i64]),
        "tgnn_sublayout_compact": (C.c_int, [p, i64, p, i32, p, i64, p, i32, p, i64, p, p, p, p, p, p, p, p, sz, p]),
        "tgnn_shard_alive_rows": (C.c_int, [p, p, i64, i64, p, i64, p, i64, p, p, p]),
        "tgnn_greedy_round_workspace_bytes": (sz, [i64]),
        "tgnn_greedy_round": (C.c_int, [p, i64, p, i64, p, i64, i32, C.c_uint64, p, p, p, p, p, p, sz, p]),
        "tgnn_greedy_finish_max_nodes": (i64, []),
        "tgnn_greedy_finish": (C.c_int, [p, i64, p, i64, i32, i32, C.c_uint64, p, p, p, p, p, p, p]),
        "tgnn_unsupervised_loss_workspace_bytes": (sz, [i32]),
        "tgnn_unsupervised_loss": (C.c_int, [p, i64, i32, p, i64, i64, p, i64, p, i64, p, i64, f32, f32, f32, p, p, p, sz, p]),
        "tgnn_solution_score_sums": (C.c_int, [p, p, i64, p, i64, p, i64, p, i64, p, p, sz, p]),
        "tgnn_graph_prep_small_max_nodes": (i64, []),
        "tgnn_graph_prep_small_max_edges": (i64, []),
        "tgnn_graph_prep_small_tmp_ints": (sz, [i64, i64, i64]),
        "tgnn_graph_prep_small": (C.c_int, [p, i64, p, i32, p, i64, i64] + [p] * 16 + [p]),
        "tgnn_graph_prep_workspace_bytes": (sz, [i64, i64, i64, i32]),
        "tgnn_graph_prep": (C.c_int, [p, i64, p, i32, p, i64, i64, i64] + [p] * 17 + [sz, p, p, p]),
        "tgnn_graph_prep_wait": (C.c_int, [p]),
        "tgnn_set_nnconv_eg": (i32, [i32]),
        "tgnn_set_dense_rows_mode": (i32, [i32]),
        "tgnn_set_lean_head": (i32, [i32]),
        "tgnn_set_prep_words_poll": (i32, [i32]),
        "tgnn_set_small_layout_limit": (None, [i64]),
        "tgnn_get_small_layout_limit": (i64, []),
        "tgnn_set_split_precision": (i32, [i32]),
        "tgnn_set_gin_fused": (i32, [i32]),
        "tgnn_set_gin_mlp_f16": (i32, [i32]),
        "tgnn_set_mid_tail": (i32, [i32]),
        "tgnn_rccl_available": (i32, []),
        "tgnn_rccl_unique_id_bytes": (sz, []),
        "tgnn_rccl_unique_id": (C.c_int, [p]),
        "tgnn_rccl_comm_create": (C.c_int, [p, i32, i32, C.POINTER(C.c_void_p)]),
        "tgnn_rccl_comm_destroy": (C.c_int, [p]),
        "tgnn_rccl_counters": (None, [p]),
        "tgnn_forward_profiled": (C.c_int, [C.POINTER(ModelDims), C.POINTER(C.c_void_p), p, p, C.POINTER(Graph), i32,
                                            i32, p, p, sz, p, C.POINTER(C.c_float), pi32]),
        "tgnn_forward_many": (C.c_int, [C.POINTER(ModelDims), C.POINTER(C.c_void_p), i32, p, p, p, i32, i32, p, p, p, p, i32, p]),
        "tgnn_forward_stamped": (C.c_int, [C.POINTER(ModelDims), C.POINTER(C.c_void_p), p, p, C.POINTER(Graph), i32, p, p, sz, p, p,
                                           C.POINTER(C.c_float)]),
        "tgnn_forward_profiled_two_stream": (C.c_int, [C.POINTER(ModelDims), C.POINTER(C.c_void_p), p, p, C.POINTER(Graph), i32,
                                                       p, p, sz, p, p, C.POINTER(C.c_float), pi32]),
        "tgnn_transpose": (C.c_int, [p, i32, i32, p, p]),
        "tgnn_swap_leading": (C.c_int, [p, i32, i32, i32, p, i32, p]),
        "tgnn_backward_workspace_bytes": (sz, [C.POINTER(ModelDims), i64, i32]),
        "tgnn_backward": (C.c_int, [C.POINTER(ModelDims), C.POINTER(C.c_void_p), C.POINTER(C.c_void_p), p, p,
                                    C.POINTER(Graph), C.POINTER(TrainGraphDesc), C.POINTER(TrainSave), p, p, p, sz, p]),
        "tgnn_forward_train": (C.c_int, [C.POINTER(ModelDims), C.POINTER(C.c_void_p), p, p, C.POINTER(Graph),
                                         C.POINTER(TrainSave), p, p, sz, p, p]),
        "tgnn_gin_aggregate": (C.c_int, [p, i64, p, p, p, p, i64, i32, p, p]),
        "tgnn_sigmoid_bwd": (C.c_int, [p, i64, p, i64, i64, i32, p, i64, p]),
        "tgnn_add_into": (C.c_int, [p, i64, i64, i32, p, i64, p]),
        "tgnn_reduce_workspace_bytes": (sz, [i32]),
        "tgnn_colsum": (C.c_int, [p, i64, i64, i32, p, p, sz, p]),
        "tgnn_bn_bwd_reduce": (C.c_int, [p, i64, p, i64, p, i64, i32, f32, p, p, p, p, sz, p]),
        "tgnn_bn_bwd_apply": (C.c_int, [p, i64, p, i64, p, p, i64, i32, i32, p, i64, p, p, i64, p]),
        "tgnn_merge_bwd_reduce": (C.c_int, [p, i64, p, p, p, p, p, i64, i32, f32, f32, p, p, p, i64, p, p, p, p, p, p,
                                            p, sz, p]),
        "tgnn_wgrad_workspace_bytes": (sz, [i64, i32, i32]),
        "tgnn_wgrad": (C.c_int, [p, i64, p, i64, i64, i64, i32, i32, p, p, p, sz, p]),
        "tgnn_sigmoid_mlp_bwd_workspace_bytes": (sz, [i64, i32, i32, i32, i32]),
        "tgnn_sigmoid_mlp_bwd": (C.c_int, [p, i64, i32, i32, i32, i32, p, p, p, p, p, p, p, i64, p, p, p, p, p, p, p, p, sz,
                                           p]),
        "tgnn_nnconv_type_sum": (C.c_int, [p, i64, p, i64, p, p, p, p, i64, i32, i32, p, p]),
        "tgnn_csr_degree": (C.c_int, [p, i64, p, p, p]),
        "tgnn_unsupervised_loss_bwd": (C.c_int, [p, i64, p, i64, i64, p, i64, p, i64, p, i64, f32, f32, f32, p, p, p, i64,
                                                 p, sz, p]),
        "tgnn_f32_to_bf16": (C.c_int, [p, i64, p, p]),
        "tgnn_nnconv64_image_elems": (sz, [i32]),
        "tgnn_nnconv64_bf16_fwd": (C.c_int, [p, i64, p, p, p, p, i32, p, p, i64, i32, p, p, p, pi32, p]),
        "tgnn_nnconv64_bf16_eg_fwd": (C.c_int, [p, i64, p, p, p, i32, p, p, i64, i32, p, p, p, pi32, p]),
        "tgnn_gin64_bf16_fwd": (C.c_int, [p, p, p, p, p, p, p, p, p, p, p, i64, i32, p, p, p, pi32, p]),
        "tgnn_collconv64_bf16_fwd": (C.c_int, [p, p, p, p, p, p, p, p, p, p, p, p, p, p, p, i64, p, p, p, p, p, p]),
        "tgnn_merge_bf16_fwd": (C.c_int, [p, p, p, p, p, i64, i32, p, p]),
        "tgnn_dense_bf16_slots_fwd": (C.c_int, [p, i64, i32, p, p, i64, i32, i32, p, p, p, pi32, p]),
        "tgnn_forward_bf16_workspace_bytes": (sz, [C.POINTER(ModelDims), i64, i32]),
        "tgnn_forward_bf16": (C.c_int, [C.POINTER(ModelDims), C.POINTER(C.c_void_p), p, p, C.POINTER(Graph), i32, p, p, sz, p, p]),
        "tgnn_forward_bf16_begin": (C.c_int, [C.POINTER(ModelDims), C.POINTER(C.c_void_p), p, i64, i32, p, sz, p]),
        "tgnn_rows_gather": (C.c_int, [p, i64, p, i64, i32, p, i64, p]),
        "tgnn_rows_scatter": (C.c_int, [p, p, i64, i32, p, i64, p]),
    }
    for name, (res, args) in sigs.items():
        fn = getattr(lib, name)      # AttributeError here = header / library mismatch
        fn.restype, fn.argtypes = res, args
    # test / experiment hooks: exported by libtgnn_debug.so only (make -C tilingnn_amd/csrc debug; TGNN_LIB_PATH selects it)
    for name, (res, args) in {"tgnn_debug_spin_fault": (None, [i32]), "tgnn_debug_set_csr_bucket_cap": (i32, [i32]),
                              "tgnn_debug_set_block_caps": (None, [i32, i32])}.items():
        try:
            fn = getattr(lib, name)
        except AttributeError:
            continue
        fn.restype, fn.argtypes = res, args
    return lib


DEBUG_LIB_PATH = os.path.join(_HERE, "libtgnn_debug.so")


def has_debug_hooks() -> bool:
    """True inside the debug build of the library (the tgnn_debug_* test hooks exist)."""
    return hasattr(lib, "tgnn_debug_spin_fault")


lib = _load()
EXPORTED_SYMBOLS = (
    "tgnn_version", "tgnn_last_error", "tgnn_csr_workspace_bytes", "tgnn_csr_build",
    "tgnn_edge_dedup_workspace_bytes", "tgnn_edge_type_dedup", "tgnn_gather_i32", "tgnn_edge_weight_table",
    "tgnn_nnconv_mean_fwd", "tgnn_nnconv_cols_max_columns", "tgnn_nnconv_cols_workspace_bytes",
    "tgnn_nnconv_cols_build", "tgnn_nnconv_cols_max_types", "tgnn_nnconv_weight_image_floats", "tgnn_nnconv_mean_cols_fwd", "tgnn_nnconv_mean_cols_f16_fwd", "tgnn_nnconv_eg_max_groups", "tgnn_nnconv_eg_build", "tgnn_nnconv_mean_eg_fwd",
    "tgnn_ubench_row_gather", "tgnn_mid_entries_words", "tgnn_mid_entries_build", "tgnn_forward_path_counts", "tgnn_set_mid_layout_limit", "tgnn_get_mid_layout_limit", "tgnn_mid_layout_max_nodes",
    "tgnn_spin_error_poll", "tgnn_set_spin_budget_us", "tgnn_persist_fallback", "tgnn_spin_error_peek", "tgnn_gin_fwd", "tgnn_dense_act_fwd", "tgnn_dense_act_slots_fwd", "tgnn_dense_act_slots_f16_fwd", "tgnn_bn_finalize", "tgnn_bn_apply",
    "tgnn_merge_fwd", "tgnn_param_count", "tgnn_param_name", "tgnn_forward_workspace_bytes", "tgnn_forward", "tgnn_forward_begin", "tgnn_forward_resume",
    "tgnn_forward_profiled", "tgnn_forward_profiled_two_stream", "tgnn_forward_stamped", "tgnn_forward_many", "tgnn_graph_prep_small_max_nodes", "tgnn_graph_prep_small_max_edges", "tgnn_graph_prep_small_tmp_ints",
    "tgnn_graph_prep_small", "tgnn_graph_prep_workspace_bytes", "tgnn_graph_prep", "tgnn_graph_prep_wait", "tgnn_set_small_layout_limit", "tgnn_get_small_layout_limit", "tgnn_set_split_precision", "tgnn_set_gin_fused", "tgnn_set_gin_mlp_f16", "tgnn_set_mid_tail", "tgnn_set_nnconv_eg", "tgnn_set_dense_rows_mode", "tgnn_set_lean_head", "tgnn_set_prep_words_poll", "tgnn_forward_begin_weights", "tgnn_forward_bf16_begin", "tgnn_forward_small_prepass", "tgnn_rccl_available", "tgnn_rccl_unique_id_bytes", "tgnn_rccl_unique_id", "tgnn_rccl_comm_create", "tgnn_rccl_comm_destroy", "tgnn_rccl_counters", "tgnn_forward_train", "tgnn_backward_workspace_bytes", "tgnn_backward", "tgnn_forward_sharded_workspace_bytes", "tgnn_forward_sharded",
    "tgnn_rows_gather", "tgnn_rows_scatter", "tgnn_unsupervised_loss_workspace_bytes", "tgnn_unsupervised_loss", "tgnn_solution_score_sums",
    "tgnn_sublayout_workspace_bytes", "tgnn_sublayout_compact", "tgnn_greedy_round_workspace_bytes", "tgnn_greedy_round", "tgnn_greedy_finish_max_nodes", "tgnn_greedy_finish", "tgnn_shard_alive_rows",
    "tgnn_transpose", "tgnn_swap_leading", "tgnn_gin_aggregate", "tgnn_sigmoid_bwd", "tgnn_add_into", "tgnn_reduce_workspace_bytes", "tgnn_colsum",
    "tgnn_bn_bwd_reduce", "tgnn_bn_bwd_apply", "tgnn_merge_bwd_reduce", "tgnn_wgrad_workspace_bytes", "tgnn_wgrad",
    "tgnn_sigmoid_mlp_bwd_workspace_bytes", "tgnn_sigmoid_mlp_bwd",
    "tgnn_nnconv_type_sum", "tgnn_csr_degree", "tgnn_unsupervised_loss_bwd",
    "tgnn_f32_to_bf16", "tgnn_nnconv64_image_elems", "tgnn_nnconv64_bf16_fwd", "tgnn_nnconv64_bf16_eg_fwd", "tgnn_gin64_bf16_fwd", "tgnn_collconv64_bf16_fwd", "tgnn_merge_bf16_fwd",
    "tgnn_dense_bf16_slots_fwd", "tgnn_forward_bf16_workspace_bytes", "tgnn_forward_bf16")


def forward_path_counts():
    """(general schedule, small-layout kernel, mid-size kernel): forwards queued so far by this process."""
    out = (C.c_int64 * 3)()
    lib.tgnn_forward_path_counts(out)
    return tuple(int(v) for v in out)


def check(rc: int) -> None:
    if rc != 0:
        msg = lib.tgnn_last_error()
        err = TgnnError(f"libtgnn error {rc}: {msg.decode() if msg else '?'}")
        err.code = int(rc)
        raise err


def ptr(t) -> C.c_void_p:
    """Device pointer of a torch tensor (None -> NULL)."""
    return C.c_void_p(0) if t is None else C.c_void_p(t.data_ptr())


import threading

_pinned = threading.local()      # .value = (device index, c_void_p) while a `pinned_stream` block is active in this thread


def current_stream(device) -> C.c_void_p:
    import torch
    hit = getattr(_pinned, "value", None)
    if hit is not None:
        return hit[1]
    return C.c_void_p(torch.cuda.current_stream(device).cuda_stream)


class pinned_stream:
    """Looks the current stream of `device` up ONCE for a block of library calls (the training step issues ~1 500 of
    them; `torch.cuda.current_stream` costs ~3 us a call).  The block must not switch streams or devices."""

    def __init__(self, device):
        self.device = device

    def __enter__(self):
        import torch
        self.prev = getattr(_pinned, "value", None)
        dev = torch.device(self.device)
        _pinned.value = (dev.index, C.c_void_p(torch.cuda.current_stream(dev).cuda_stream))

    def __exit__(self, *exc):
        _pinned.value = self.prev


_side_streams = {}
_stream_lock = threading.Lock()
_stream_retry_at = {}     # device index -> monotonic time before which a short search is not repeated
_stream_sets = {}         # device index -> streams found to run beside the device's current stream and beside each other


def _run_side_by_side(a, b, dev, cycles=400_000, reps=3) -> bool:
    """True when a spinning kernel on stream a and one on stream b overlap in time.  Timed on the DEVICE (events on the two
    streams; the minimum of a few repetitions), so that a busy host -- a loaded box, a Python thread holding the GIL between the
    two launches -- cannot make two concurrent queues look like colliding ones."""
    import torch

    def span(with_b: bool) -> float:
        best = float("inf")
        for _ in range(reps):
            torch.cuda.synchronize(dev)
            e0, ea, eb = (torch.cuda.Event(enable_timing=True) for _ in range(3))
            e0.record(a)
            if with_b:
                b.wait_event(e0)                             # both spins start behind the same point of stream a
            with torch.cuda.stream(a):
                torch.cuda._sleep(cycles)
            ea.record(a)
            if with_b:
                with torch.cuda.stream(b):
                    torch.cuda._sleep(cycles)
                eb.record(b)
            torch.cuda.synchronize(dev)
            t = e0.elapsed_time(ea)
            if with_b:
                t = max(t, e0.elapsed_time(eb))
            best = min(best, t)
        return best
    span(False)                                             # (warm-up: first use of a stream binds it to a hardware queue)
    with torch.cuda.stream(b):
        torch.cuda._sleep(1000)
    return span(True) < 1.5 * span(False)


def concurrent_streams(device, k: int):
    """k streams of this package's own that really run beside the device's current stream and beside each other.  HIP binds
    streams to its hardware queues (4 by default, GPU_MAX_HW_QUEUES) round robin, and two streams on one queue run their kernels
    strictly one after the other -- which streams collide depends on how many the process (torch, RCCL, ...) created before.
    Measured, not assumed: candidates are created one at a time and kept when a spinning kernel on them overlaps with one on the
    current stream and on every stream kept so far (~1 ms per test, once per device).  Fewer than k found: the last ones repeat."""
    import torch
    dev = torch.device(device)
    key = dev.index if dev.index is not None else torch.cuda.current_device()
    with _stream_lock:
        return _concurrent_streams_locked(key, k)


def _concurrent_streams_locked(key, k):
    import torch
    import time
    have = _stream_sets.setdefault(key, [])
    # (a search that came up short is repeated after a minute, not never: what else the process runs changes)
    if len(have) < k and time.monotonic() >= _stream_retry_at.get(key, 0.0):
        cur = torch.cuda.current_stream(key)
        tries = 0
        while len(have) < k and tries < 16:
            tries += 1
            cand = torch.cuda.Stream(device=key)
            if _run_side_by_side(cur, cand, key) and all(_run_side_by_side(h, cand, key) for h in have):
                have.append(cand)
        if len(have) < k:
            _stream_retry_at[key] = time.monotonic() + 60.0
            if not have:
                have.append(torch.cuda.Stream(device=key))
    return [have[min(i, len(have) - 1)] for i in range(k)]


def remeasure_side_streams(device=None) -> None:
    """Forget the side streams picked so far (of one device, or of all): the next forward measures again."""
    import torch
    with _stream_lock:
        keys = list(_stream_sets) if device is None else [torch.device(device).index if torch.device(device).index is not None
                                                          else torch.cuda.current_device()]
        for key in keys:
            _stream_sets.pop(key, None)
            _side_streams.pop(key, None)
            _stream_retry_at.pop(key, None)


def side_stream(device) -> C.c_void_p:
    """Second stream of the forward's two-chain schedule (the collision branch runs free beside the adjacency branch,
    csrc/forward.hip); one per device, picked on first use among streams that were MEASURED to run beside the current one
    (concurrent_streams).  TGNN_TWO_STREAMS=0 returns NULL = everything on the current stream."""
    import torch
    if os.environ.get("TGNN_TWO_STREAMS", "1") == "0":
        return C.c_void_p(None)
    key = torch.device(device).index
    if key is None:
        key = torch.cuda.current_device()
    st = _side_streams.get(key)
    if st is None:
        st = _side_streams[key] = concurrent_streams(key, 1)[0]
    return C.c_void_p(st.cuda_stream)


ERR_UNSUPPORTED = -4          # TGNN_ERR_UNSUPPORTED (include/tgnn.h)
ERR_STALE_RESULT = -6         # TGNN_ERR_STALE_RESULT
ERR_UNVERIFIED = -7           # TGNN_ERR_UNVERIFIED


def side_stream_torch(device):
    """The torch.cuda.Stream behind side_stream(device) (None when the two-chain schedule is off)."""
    import torch
    if not side_stream(device).value:
        return None
    key = torch.device(device).index
    return _side_streams.get(torch.cuda.current_device() if key is None else key)


def param_names(dims: ModelDims):
    n = lib.tgnn_param_count(C.byref(dims))
    if n < 0:
        raise TgnnError("invalid model dims")
    buf = C.create_string_buffer(256)
    out = []
    for i in range(n):
        check(lib.tgnn_param_name(C.byref(dims), i, buf, 256))
        out.append(buf.value.decode())
    return out
