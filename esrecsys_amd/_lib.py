"""ctypes binding of libesr_hip.so (include/esr_hip.h).

The library is the ONLY compute backend: there is no CPU or eager-PyTorch fallback.  If the
shared object is missing or fails to load, every op raises ``EsrLibraryError``.
"""
import ctypes
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libesr_hip.so")

ESR_OK = 0
ESR_F32 = 0
ESR_BF16 = 1
ESR_PROBE_LIVE_DATA = 0x100
ESR_PROBE_F16 = 2  # esr_probe_mfma: dtype | this = full-entropy operands
GLOVE_REFERENCE = 0
GLOVE_DIAGONAL = 1
RETRIEVE_EXACT = 0
RETRIEVE_BF16 = 1
RETRIEVE_F16X2 = 2
RETRIEVE_F16R = 3

c_i32p = ctypes.c_void_p
c_f32p = ctypes.c_void_p
c_vp = ctypes.c_void_p
c_i64 = ctypes.c_int64
c_int = ctypes.c_int
c_int32 = ctypes.c_int32
c_uint32 = ctypes.c_uint32
c_f32 = ctypes.c_float
c_size = ctypes.c_size_t

# name -> (restype, argtypes); mirrors include/esr_hip.h one for one.
SIGNATURES = {
    "esr_last_error": (ctypes.c_char_p, []),
    "esr_version": (c_int, []),
    "esr_device_info": (c_int, [ctypes.POINTER(c_int), ctypes.POINTER(c_int), ctypes.POINTER(c_size),
                                ctypes.c_char_p, c_int]),
    "esr_kernel_timing": (c_int, [c_int]),
    "esr_kernel_timing_read": (ctypes.c_long, [ctypes.c_char_p, c_size]),
    "esr_trace_markers": (c_int, [c_int]),
    "esr_gather_rows": (c_int, [c_vp, c_int, c_i64, c_int, c_i32p, c_i64, c_vp, c_vp]),
    "esr_check_ids": (c_int, [c_i32p, c_i64, c_i64, c_vp, c_vp]),
    "esr_unpermute_rows": (c_int, [c_vp, c_int, c_int, c_i32p, c_i64, c_vp, c_vp]),
    "esr_unpermute_rows_bf16_to_f32": (c_int, [c_vp, c_int, c_i32p, c_i64, c_f32p, c_vp]),
    "esr_glove_forward": (c_int, [c_f32p, c_f32p, c_i64, c_int, c_i32p, c_i64, c_f32p, c_f32p, c_vp]),
    "esr_glove_workspace_bytes": (c_size, [c_i64]),
    "esr_glove_fwd_bwd": (c_int, [c_f32p, c_f32p, c_i64, c_int, c_i32p, c_f32p, c_i64, c_int, c_f32p, c_f32p,
                                  c_f32p, c_vp, c_size, c_vp]),
    "esr_glove_step_workspace_bytes": (c_size, [c_i64, c_int]),
    "esr_glove_train_step": (c_int, [c_vp, c_vp, c_vp, c_f32p, c_f32p, c_f32p, c_i64, c_int, c_int, c_i32p, c_f32p, c_i64,
                                     c_int, c_f32, c_f32, c_uint32, c_i32p, c_i32p, c_vp, c_int, c_int, c_vp, c_uint32, c_f32p,
                                     c_vp, c_size, c_vp]),
    "esr_stream_gate": (c_int, [c_vp, c_uint32, c_uint32, c_vp]),
    "esr_glove_train_steps": (c_int, [c_vp, c_vp, c_vp, c_f32p, c_f32p, c_f32p, c_i64, c_int, c_int, c_int, c_vp, c_vp, c_i64,
                                      c_int, c_f32, c_f32, c_uint32, c_i32p, c_i32p, c_vp, c_vp, c_f32p, c_vp, c_size,
                                      c_vp]),
    "esr_glove_plan_bytes": (c_size, [c_i64]),
    "esr_glove_plan": (c_int, [c_vp, c_vp, c_int, c_i64, c_i32p, c_i32p, c_vp, c_vp, c_int32, c_vp]),
    "esr_segment_sum_rows": (c_int, [c_f32p, c_i64, c_int, c_i32p, c_i32p, c_i64, c_f32p, c_vp]),
    "esr_unique_by_owner_workspace_bytes": (c_size, [c_i64]),
    "esr_unique_by_owner": (c_int, [c_vp, c_vp, c_vp, c_int, c_int, c_i64, c_i32p, c_i32p, c_i32p, c_i32p, c_vp, c_vp,
                                    c_size, c_vp]),
    "esr_topk_columns": (c_int, [c_f32p, c_i64, c_int, c_int, c_f32p, c_i32p, c_vp]),
    "esr_momentum_catchup_rows": (c_int, [c_f32p, c_f32p, c_i32p, c_i64, c_int, c_i32p, c_i64, c_int, c_int, c_f32, c_f32,
                                          c_vp]),
    "esr_sparse_momentum_step": (c_int, [c_f32p, c_f32p, c_i64, c_int, c_i32p, c_i32p, c_i64, c_f32p, c_f32, c_f32, c_vp]),
    "esr_momentum_flush": (c_int, [c_f32p, c_f32p, c_i32p, c_i64, c_int, c_int, c_f32, c_f32, c_vp]),
    "esr_momentum_catchup_rows2": (c_int, [c_f32p, c_f32p, c_i32p, c_i32p, c_int, c_f32p, c_f32p, c_i32p, c_i32p, c_int, c_int,
                                           c_i64, c_int, c_f32, c_f32, c_vp]),
    "esr_sparse_momentum_step_multi": (c_int, [c_vp, c_vp, c_vp, c_int, c_int, c_i32p, c_i32p, c_i64, c_f32p, c_f32, c_f32,
                                               c_vp]),
    "esr_spotify_train_step_workspace_bytes": (c_size, [c_int, c_int, c_int, c_int]),
    "esr_spotify_train_step": (c_int, [c_f32p, c_f32p, c_i32p, c_i64, c_f32p, c_f32p, c_i32p, c_i64, c_int, c_i32p, c_i32p,
                                       c_int, c_int, c_int, c_f32, c_int, c_f32, c_f32, c_f32p, c_vp, c_size, c_vp]),
    "esr_ivf_search_workspace_bytes": (c_size, [c_i64, c_int, c_int, c_int, c_int]),
    "esr_ivf_search": (c_int, [c_f32p, c_i64, c_int, c_f32p, c_i32p, c_i32p, c_int, c_int, c_i32p, c_int, c_int, c_f32p,
                               c_i32p, c_vp, c_size, c_vp]),
    "esr_long_run_hint": (c_int, [c_i32p, c_i64, c_int, c_vp, c_int32, c_vp]),
    "esr_rows_consolidate": (c_int, [c_vp, c_vp, c_vp, c_i64, c_int, c_int, c_vp]),
    "esr_rows_restamp": (c_int, [c_vp, c_i64, c_vp]),
    "esr_triplet_workspace_bytes": (c_size, [c_i64]),
    "esr_triplet_fwd_bwd": (c_int, [c_f32p, c_i64, c_f32p, c_i64, c_f32p, c_i64, c_int, c_i32p, c_i32p, c_i32p, c_i64, c_f32, c_f32,
                                    c_int, c_f32p, c_f32p, c_f32p, c_f32p, c_f32p, c_f32p, c_vp, c_size, c_vp]),
    "esr_triplet_step_workspace_bytes": (c_size, [c_i64, c_int]),
    "esr_triplet_train_step": (c_int, [c_vp, c_vp, c_vp, c_f32p, c_i64, c_vp, c_vp, c_vp, c_f32p, c_i64, c_int, c_int,
                                       c_i32p, c_i32p, c_i32p, c_i64, c_f32, c_f32, c_f32, c_f32, c_uint32, c_i32p,
                                       c_i32p, c_vp, c_int, c_f32p, c_vp, c_size, c_vp]),
    "esr_triplet_train_steps": (c_int, [c_vp, c_vp, c_vp, c_f32p, c_i64, c_vp, c_vp, c_vp, c_f32p, c_i64, c_int, c_int,
                                        c_int, c_vp, c_i64, c_f32, c_f32, c_f32, c_f32, c_uint32, c_i32p, c_i32p, c_vp,
                                        c_vp, c_f32p, c_vp, c_size, c_vp]),
    "esr_triplet_plan_bytes": (c_size, [c_i64]),
    "esr_triplet_plan": (c_int, [c_vp, c_int, c_i64, c_i64, c_i32p, c_i32p, c_vp, c_vp, c_int32, c_vp]),
    "esr_inbatch_workspace_bytes": (c_size, [c_i64, c_int]),
    "esr_inbatch_softmax_fwd_bwd": (c_int, [c_f32p, c_f32p, c_i64, c_int, c_f32, c_f32, c_f32, c_f32p, c_f32p, c_f32p,
                                            c_f32p, c_vp, c_size, c_vp]),
    "esr_inbatch3_workspace_bytes": (c_size, [c_i64, c_int]),
    "esr_inbatch_softmax_fwd_bwd_bf16x3": (c_int, [c_f32p, c_f32p, c_i64, c_int, c_f32, c_f32, c_f32, c_f32p, c_f32p,
                                                   c_f32p, c_f32p, c_vp, c_size, c_vp]),
    "esr_inbatch_towers_fwd_bwd_bf16x3": (c_int, [c_vp, c_i64, c_vp, c_i64, c_int, c_int, c_i32p, c_i32p, c_i32p, c_i32p,
                                                  c_i64, c_f32, c_f32, c_f32, c_f32p, c_f32p, c_f32p, c_f32p, c_vp,
                                                  c_size, c_vp]),
    "esr_inbatch2h_workspace_bytes": (c_size, [c_i64, c_int]),
    "esr_inbatch_softmax_fwd_bwd_f16x2": (c_int, [c_f32p, c_f32p, c_i64, c_int, c_f32, c_f32, c_f32, c_f32p, c_f32p,
                                                  c_f32p, c_f32p, c_vp, c_size, c_vp]),
    "esr_inbatch_towers_fwd_bwd_f16x2": (c_int, [c_vp, c_i64, c_vp, c_i64, c_int, c_int, c_i32p, c_i32p, c_i32p, c_i32p,
                                                 c_i64, c_f32, c_f32, c_f32, c_f32p, c_f32p, c_f32p, c_f32p, c_vp,
                                                 c_size, c_vp]),
    "esr_inbatch2h_pass_c_forms": (c_int, [c_vp, c_size, c_i64, c_i32p, c_vp]),
    "esr_inbatch_train_step_workspace_bytes": (c_size, [c_i64, c_int]),
    "esr_inbatch_train_step_f16x2": (c_int, [c_vp, c_f32p, c_i64, c_vp, c_f32p, c_i64, c_int, c_int, c_i32p, c_i32p, c_i64,
                                             c_f32, c_f32, c_f32, c_f32, c_f32, c_i32p, c_i32p, c_int, c_f32p, c_f32p,
                                             c_vp, c_size, c_vp, c_vp]),
    "esr_segment_sort_workspace_bytes": (c_size, [c_i64]),
    "esr_segment_sort_ids": (c_int, [c_i32p, c_i64, c_i64, c_i32p, c_i32p, c_vp, c_size, c_vp]),
    "esr_segment_sort_ids_multi": (c_int, [c_vp, c_vp, c_vp, c_int, c_i64, c_i32p, c_i32p, c_vp, c_size, c_vp]),
    "esr_segment_sort_batched_workspace_bytes": (c_size, [c_i64, c_int]),
    "esr_segment_sort_ids_batched": (c_int, [c_vp, c_vp, c_vp, c_int, c_int, c_i64, c_i32p, c_i32p, c_vp, c_size, c_vp]),
    "esr_sparse_adagrad_scatter": (c_int, [c_vp, c_int, c_f32p, c_i64, c_int, c_i32p, c_i32p, c_i64, c_f32p, c_f32,
                                           c_f32, c_vp]),
    "esr_concat_offset_ids": (c_int, [c_vp, c_vp, c_vp, c_int, c_i32p, c_vp]),
    "esr_gather_rows_multi": (c_int, [c_vp, c_vp, c_int, c_int, c_int, c_i32p, c_i64, c_vp, c_vp]),
    "esr_sparse_adagrad_scatter_multi": (c_int, [c_vp, c_vp, c_vp, c_int, c_int, c_int, c_i32p, c_i32p, c_i64, c_f32p,
                                                 c_f32, c_f32, c_int, c_vp]),
    "esr_sparse_sgd_scatter": (c_int, [c_vp, c_int, c_i64, c_int, c_i32p, c_i32p, c_i64, c_f32p, c_f32, c_vp]),
    "esr_rows_to_dense": (c_int, [c_f32p, c_i64, c_int, c_i32p, c_i32p, c_i64, c_f32p, c_vp]),
    "esr_dense_adam": (c_int, [c_f32p, c_f32p, c_f32p, c_f32p, c_i64, c_f32, c_f32, c_f32, c_f32, c_i64, c_vp]),
    "esr_score_all": (c_int, [c_f32p, c_i64, c_int, c_i32p, c_int, c_f32p, c_vp]),
    "esr_argsort_columns_workspace_bytes": (c_size, [c_i64, c_int]),
    "esr_argsort_columns": (c_int, [c_f32p, c_i64, c_int, c_i32p, c_vp, c_size, c_vp]),
    "esr_score_topk_workspace_bytes": (c_size, [c_i64, c_i64, c_int]),
    "esr_score_topk": (c_int, [c_f32p, c_f32p, c_i64, c_i64, c_int, c_int, c_f32p, c_i32p, c_vp, c_size, c_vp]),
    "esr_retrieve_workspace_bytes": (c_size, [c_i64, c_i64, c_int, c_int, c_int]),
    "esr_retrieve_topk": (c_int, [c_f32p, c_f32p, c_i64, c_i64, c_int, c_int, c_int, ctypes.c_int32, ctypes.c_int32,
                                  c_f32p, c_i32p, c_vp, c_size, c_vp]),
    "esr_retrieve_prepared_bytes": (c_size, [c_i64, c_int, c_int]),
    "esr_retrieve_prepare": (c_int, [c_f32p, c_i64, c_int, c_int, c_vp, c_size, c_vp]),
    "esr_retrieve_topk_prepared": (c_int, [c_f32p, c_f32p, c_vp, c_i64, c_i64, c_int, c_int, c_int, ctypes.c_int32,
                                           ctypes.c_int32, c_f32p, c_i32p, c_vp, c_size, c_vp]),
    "esr_rescore_candidates": (c_int, [c_f32p, c_f32p, c_i64, c_i64, c_int, c_i32p, c_int, ctypes.c_int32,
                                       ctypes.c_int32, c_f32p, c_vp]),
    "esr_topk_merge": (c_int, [c_f32p, c_i32p, c_i64, c_int, c_int, c_f32p, c_i32p, c_vp]),
    "esr_recall_at_k": (c_int, [c_i32p, c_i64, c_int, c_i32p, c_int, c_vp, c_vp]),
    "esr_run_offsets": (c_int, [c_i32p, c_i64, c_int, c_i32p, c_i32p, c_vp]),
    "esr_ivf_centroids": (c_int, [c_f32p, c_i32p, c_f32p, c_i32p, c_int, c_int, c_f32p, c_vp]),
    "esr_sorted_membership": (c_int, [c_i32p, c_i64, c_i32p, c_i64, c_int, c_int, c_vp, c_vp]),
    "esr_flagged_first_workspace_bytes": (c_size, [c_i64, c_int]),
    "esr_flagged_first": (c_int, [c_vp, c_i32p, c_i64, c_int, c_vp, c_int, c_i32p, c_vp, c_i64, c_i64, c_vp, c_size, c_vp]),
    "esr_spotify_workspace_bytes": (c_size, [c_int, c_int, c_int, c_int]),
    "esr_spotify_get_embeddings": (c_int, [c_f32p, c_i64, c_f32p, c_i64, c_int, c_i32p, c_i32p, c_i64, c_f32p, c_f32p,
                                           c_vp]),
    "esr_spotify_forward": (c_int, [c_f32p, c_i64, c_f32p, c_i64, c_int, c_i32p, c_i32p, c_int, c_int, c_int, c_f32p,
                                    c_f32p, c_f32p, c_f32p, c_f32p, c_f32p, c_vp, c_size, c_vp]),
    "esr_spotify_fwd_bwd": (c_int, [c_f32p, c_i64, c_f32p, c_i64, c_int, c_i32p, c_i32p, c_int, c_int, c_int, c_f32,
                                    c_f32p, c_i32p, c_f32p, c_f32p, c_vp, c_size, c_vp]),
    "esr_spotify_affinity_all": (c_int, [c_f32p, c_i64, c_f32p, c_i64, c_int, c_i32p, c_i32p, c_int, c_i32p, c_i32p,
                                         c_i64, c_f32p, c_vp]),
    "esr_dense_momentum_decay": (c_int, [c_f32p, c_f32p, c_i64, c_f32, c_f32, c_vp]),
    "esr_sparse_momentum_scatter": (c_int, [c_f32p, c_f32p, c_i64, c_int, c_i32p, c_i32p, c_i64, c_f32p, c_f32,
                                            c_vp]),
    "esr_bucket_workspace_bytes": (c_size, [c_i64]),
    "esr_bucket_ids_by_owner": (c_int, [c_i32p, c_i64, c_int, c_i32p, c_i32p, c_i32p, c_vp, c_vp, c_size, c_vp]),
    "esr_bucket_ids_by_owner_multi": (c_int, [c_vp, c_vp, c_vp, c_int, c_int, c_i32p, c_i32p, c_i32p, c_vp, c_vp, c_size,
                                              c_vp]),
    "esr_bucket_batched_workspace_bytes": (c_size, [c_i64, c_int]),
    "esr_bucket_ids_by_owner_batched": (c_int, [c_vp, c_vp, c_vp, c_int, c_int, c_int, c_i32p, c_i32p, c_i32p, c_vp, c_vp,
                                                c_size, c_vp]),
    "esr_comm_load": (c_int, [ctypes.c_char_p]),
    "esr_comm_unique_id": (c_int, [c_vp]),
    "esr_comm_init": (c_int, [c_vp, c_int, c_int, ctypes.POINTER(c_vp)]),
    "esr_comm_count": (c_int, [c_vp, ctypes.POINTER(c_int), ctypes.POINTER(c_int)]),
    "esr_comm_async_error": (c_int, [c_vp]),
    "esr_comm_abort": (c_int, [c_vp]),
    "esr_comm_destroy": (c_int, [c_vp]),
    "esr_alltoall_bytes": (c_int, [c_vp, c_vp, c_vp, c_vp, c_vp, c_vp]),
    "esr_alltoall_bytes_multi": (c_int, [c_vp, c_int, c_vp, c_vp, c_vp, c_vp, c_vp]),
    "esr_allgather_bytes": (c_int, [c_vp, c_vp, c_i64, c_vp, c_vp]),
    "esr_alltoall_ids": (c_int, [c_vp, c_i32p, c_vp, c_i32p, c_vp, c_vp]),
    "esr_alltoall_rows": (c_int, [c_vp, c_vp, c_int, c_int, c_vp, c_vp, c_vp, c_vp]),
    "esr_alltoall_grads": (c_int, [c_vp, c_f32p, c_int, c_vp, c_f32p, c_vp, c_vp]),
    "esr_rows_f32_to_bf16": (c_int, [c_f32p, c_i64, c_int, c_vp, c_vp]),
    "esr_sharded_lookup": (c_int, [c_vp, c_int, c_vp, c_vp, c_int, c_int, c_int, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp]),
    "esr_sharded_update": (c_int, [c_vp, c_int, c_vp, c_vp, c_vp, c_int, c_int, c_int, c_vp, c_i64, c_vp, c_vp, c_vp, c_vp,
                                   c_vp, c_int, c_vp, c_vp, c_vp, c_vp, c_vp, c_f32, c_f32, c_int, c_vp]),
    "esr_sharded_triplet_step_workspace_bytes": (c_size, [c_vp, c_vp, c_i64]),
    "esr_sharded_triplet_step": (c_int, [c_vp, c_vp, c_i64, c_f32, c_f32, c_f32, c_f32, c_vp, c_vp, c_size, c_vp]),
    "esr_sharded_glove_step_workspace_bytes": (c_size, [c_vp, c_vp, c_vp, c_i64]),
    "esr_sharded_glove_step": (c_int, [c_vp, c_vp, c_vp, c_vp, c_i64, c_int, c_f32, c_f32, c_vp, c_vp, c_size, c_vp]),
    "esr_sharded_step_overlap_workspace_bytes": (c_size, [c_vp, c_vp]),
    "esr_sharded_overlap_release": (None, [c_vp]),
    "esr_sharded_triplet_step_overlapped": (c_int, [c_vp, c_vp, c_vp, c_i64, c_f32, c_f32, c_f32, c_f32, c_vp, c_vp, c_size,
                                                    c_vp]),
    "esr_sharded_glove_step_overlapped": (c_int, [c_vp, c_vp, c_vp, c_vp, c_vp, c_i64, c_int, c_f32, c_f32, c_vp, c_vp,
                                                  c_size, c_vp]),
}


class ShardGroupStruct(ctypes.Structure):
    """esr_shard_group_t (include/esr_hip.h)"""
    _fields_ = [("comm", ctypes.c_void_p), ("world", ctypes.c_int), ("tables", ctypes.c_void_p),
                ("accums", ctypes.c_void_p), ("row_offsets", ctypes.c_void_p), ("ntables", ctypes.c_int),
                ("dtype", ctypes.c_int), ("D", ctypes.c_int), ("grad_dtype", ctypes.c_int)]


class RoutingPlanStruct(ctypes.Structure):
    """esr_routing_plan_t (include/esr_hip.h)"""
    _fields_ = [("asked_rows", ctypes.c_void_p), ("asked_counts", ctypes.c_void_p), ("ask_counts", ctypes.c_void_p),
                ("index", ctypes.c_void_p), ("sorted_uidx", ctypes.c_void_p), ("occ_perm", ctypes.c_void_p),
                ("owner_sorted", ctypes.c_void_p), ("owner_perm", ctypes.c_void_p), ("long_runs", ctypes.c_int)]


class StepOverlapStruct(ctypes.Structure):
    """esr_step_overlap_t (include/esr_hip.h)"""
    _fields_ = [("back", ctypes.c_void_p * 2), ("ready", ctypes.c_void_p), ("stale_rows", ctypes.c_void_p),
                ("stale_asked", ctypes.c_void_p), ("stale_pos", ctypes.c_void_p), ("stale_ask", ctypes.c_void_p),
                ("next_plan", ctypes.c_void_p), ("next_back", ctypes.c_void_p * 2), ("next_served", ctypes.c_void_p * 2),
                ("comm2", ctypes.c_void_p), ("side", ctypes.c_void_p), ("next_ready", ctypes.c_void_p)]


class EsrLibraryError(RuntimeError):
    """libesr_hip.so is missing / failed to load, or a call returned an ESR_E* code."""


_lib = None


def load(path=None):
    """Load libesr_hip.so (once).  torch must be imported first so that the HIP runtime the
    library binds to (soname libamdhip64.so.7) is the one torch already loaded."""
    global _lib
    if _lib is not None:
        return _lib
    path = path or os.environ.get("ESR_HIP_LIB") or LIB_PATH  # (ESR_HIP_LIB: another build of the library, for A/B runs)
    if not os.path.exists(path):
        raise EsrLibraryError(
            "%s not found: build it with `python -m esrecsys_amd.build` (hipcc, gfx950). "
            "There is no CPU fallback." % path)
    try:
        import torch  # noqa: F401  (loads torch's libamdhip64 first)
    except ImportError:
        pass
    try:
        lib = ctypes.CDLL(path, mode=ctypes.RTLD_GLOBAL)
    except OSError as e:  # pragma: no cover
        raise EsrLibraryError("failed to load %s: %s" % (path, e))
    for name, (res, args) in SIGNATURES.items():
        try:
            fn = getattr(lib, name)
        except AttributeError:
            raise EsrLibraryError("%s does not export %s (stale build?)" % (path, name))
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


# include/esr_probe.h: the measurement probes live in their own shared object (not part of the product ABI)
PROBE_LIB_PATH = os.path.join(_HERE, "libesr_probe.so")
PROBE_SIGNATURES = {
    "esr_probe_mfma": (c_int, [c_int, c_int, c_int, c_f32p, ctypes.POINTER(ctypes.c_double), c_vp]),
    "esr_probe_hbm_read": (c_int, [c_vp, c_i64, c_int, c_int, c_f32p, c_vp]),
    "esr_probe_mfma_valu": (c_int, [c_int, c_int, c_int, c_int, c_int, c_int, c_vp, c_f32p, c_vp]),
}
_probe = None


def load_probe(path=None):
    """Load libesr_probe.so (once; after libesr_hip.so, whose error helpers it resolves)."""
    global _probe
    if _probe is not None:
        return _probe
    load()
    path = path or PROBE_LIB_PATH
    if not os.path.exists(path):
        raise EsrLibraryError("%s not found: build it with `python -m esrecsys_amd.build`" % path)
    try:
        lib = ctypes.CDLL(path, mode=ctypes.RTLD_GLOBAL)
    except OSError as e:  # pragma: no cover
        raise EsrLibraryError("failed to load %s: %s" % (path, e))
    for name, (res, args) in PROBE_SIGNATURES.items():
        fn = getattr(lib, name)
        fn.restype = res
        fn.argtypes = args
    _probe = lib
    return lib


def check(rc, what):
    if rc != ESR_OK:
        msg = load().esr_last_error()
        raise EsrLibraryError("%s failed (code %d): %s" % (what, rc, msg.decode() if msg else "?"))
