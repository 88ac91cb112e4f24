"""ctypes binding of libshodh_hip.so (the C ABI declared in include/shodh_hip.h).

There is no fallback: if the HIP library is missing or fails to load, importing the product
fails loudly. The oracle (oracle/) is never imported from here.
"""
import ctypes as C
import os

# PyTorch-ROCm wheels bundle their own libamdhip64 / libhsa-runtime64 and load them by path. A
# process must not initialise two HIP runtimes (the second one finds no device), so when torch is
# installed it is imported FIRST: libshodh_hip.so (DT_NEEDED libamdhip64.so.7) then binds to the
# runtime already in the process. Without torch the library binds to /opt/rocm/lib as usual. The
# C library itself has no torch dependency.
try:
    import torch as _torch  # noqa: F401
except ImportError:          # standalone use from C/C++/Rust or a torch-free python
    _torch = None

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("SHODH_HIP_LIB") or os.path.join(HERE, "libshodh_hip.so")      # SHODH_HIP_LIB: a diagnostic build of the same library (tools/)

OK = 0
ERR_INVALID, ERR_DIM, ERR_DEVICE, ERR_OOM, ERR_STATE, ERR_IO, ERR_NONFINITE, ERR_UNSUPPORTED = -1, -2, -3, -4, -5, -6, -7, -8
METRIC_NDP, METRIC_EUCLIDEAN, METRIC_COSINE = 0, 1, 2
ORDER_SCALAR4, ORDER_AVX2, ORDER_SEQ_1M = 0, 1, 2
INDEX_FLAT, INDEX_IVFPQ = 0, 1
SCAN_AUTO, SCAN_EXACT, SCAN_MFMA, SCAN_GRAPH = 0, 1, 2, 3
DTYPE_FP32, DTYPE_BF16, DTYPE_INT8 = 0, 1, 2
QUANT_SCOPE_BATCH, QUANT_SCOPE_PER_TEXT = 0, 1      # shodh_embed_cfg.quant_scope
WEIGHT_ABSENT, WEIGHT_F32, WEIGHT_EXPORT_Q8, WEIGHT_SELF_Q8 = 0, 1, 2, 3


class ShodhError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__("shodh_hip error %d: %s" % (code, msg))
        self.code = code


class IndexCfg(C.Structure):
    _fields_ = [("dim", C.c_uint32), ("metric", C.c_uint32), ("kind", C.c_uint32), ("order", C.c_uint32),
                ("device", C.c_int32), ("scan_mode", C.c_uint32), ("reserve_rows", C.c_uint64),
                ("id_base", C.c_uint64), ("nprobe", C.c_uint32), ("reserved", C.c_uint32),
                ("max_degree", C.c_uint32), ("search_list_size", C.c_uint32), ("alpha", C.c_float), ("reserved2", C.c_uint32)]


class EmbedCfg(C.Structure):
    _fields_ = [("device", C.c_int32), ("dtype", C.c_uint32), ("max_len", C.c_uint32), ("vocab", C.c_uint32),
                ("hidden", C.c_uint32), ("layers", C.c_uint32), ("heads", C.c_uint32), ("intermediate", C.c_uint32),
                ("max_pos", C.c_uint32), ("type_vocab", C.c_uint32), ("ln_eps", C.c_float), ("compute_padded", C.c_uint32),
                ("quant_scope", C.c_uint32), ("weights_path", C.c_char_p)]


class Weights(C.Structure):
    _fields_ = [("semantic", C.c_float), ("entity", C.c_float), ("tag", C.c_float), ("importance", C.c_float),
                ("momentum", C.c_float), ("access_count", C.c_float), ("graph_strength", C.c_float),
                ("update_count", C.c_uint32)]


class LegFusionCfg(C.Structure):
    _fields_ = [("fusion_v2", C.c_uint8), ("fusion_flat", C.c_uint8), ("fusion_sum", C.c_uint8), ("fusion_rrf", C.c_uint8),
                ("isolate_leg", C.c_uint8), ("flat_adaptive", C.c_uint8), ("adapt_feature", C.c_uint8), ("adapt_symmetric", C.c_uint8),
                ("graph_w", C.c_float), ("hybrid_w", C.c_float), ("rrf_k", C.c_float), ("flat_consensus", C.c_float),
                ("adapt_trust_max", C.c_float), ("fw_graph", C.c_float), ("fw_vec", C.c_float), ("fw_bm25", C.c_float),
                ("agree_k", C.c_float), ("agree_lo", C.c_float), ("agree_hi", C.c_float), ("peak_lo", C.c_float), ("peak_hi", C.c_float)]


class RelevanceCfg(C.Structure):
    _fields_ = [("min_importance", C.c_float), ("recency_boost_hours", C.c_uint64), ("recency_boost_multiplier", C.c_float),
                ("graph_boost_multiplier", C.c_float), ("max_results", C.c_uint32)]


class ShardedCfg(C.Structure):
    _fields_ = [("dim", C.c_uint32), ("metric", C.c_uint32), ("kind", C.c_uint32), ("order", C.c_uint32), ("scan_mode", C.c_uint32),
                ("nprobe", C.c_uint32), ("block_log2", C.c_uint32), ("exchange", C.c_uint32), ("reserve_rows_per_shard", C.c_uint64)]


EXCHANGE_AUTO, EXCHANGE_RCCL, EXCHANGE_COPY = 0, 1, 2

# every symbol include/shodh_hip.h declares: name -> (restype, argtypes)
_vp, _fp, _u8p, _u32p, _u64p, _i32p = C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p
class VamaInfo(C.Structure):
    _fields_ = [("num_vectors", C.c_uint64), ("dimension", C.c_uint32), ("max_degree", C.c_uint32), ("medoid", C.c_uint32),
                ("deleted_count", C.c_uint32), ("incremental_inserts", C.c_uint64), ("graph_edges", C.c_uint64), ("distance_metric", C.c_uint8)]


class SpanInfo(C.Structure):
    _fields_ = [("num_vectors", C.c_uint64), ("total_postings", C.c_uint64), ("num_partitions", C.c_uint32), ("dimension", C.c_uint32),
                ("pq_subvectors", C.c_uint32), ("pq_num_centroids", C.c_uint32), ("pq_subvec_dim", C.c_uint32),
                ("pq_enabled", C.c_uint8), ("distance_metric", C.c_uint8)]


SYMBOLS = {
    "shodh_to_lowercase": (C.c_size_t, [C.c_char_p, C.c_char_p, C.c_size_t]),
    "shodh_guard_mode": (C.c_int, []),
    "shodh_guard_stats": (C.c_int, [C.POINTER(C.c_uint64), C.POINTER(C.c_uint64), C.POINTER(C.c_uint64)]),
    "shodh_guard_torch_alloc": (C.c_void_p, [C.c_int64, C.c_int, C.c_void_p]),
    "shodh_guard_torch_free": (None, [C.c_void_p, C.c_int64, C.c_int, C.c_void_p]),
    "shodh_last_error": (C.c_char_p, []),
    "shodh_abi_version": (C.c_int, []),
    "shodh_index_graph_overflowed": (C.c_int, [_vp]),
    "shodh_device_count": (C.c_int, []),
    "shodh_device_info": (C.c_int, [C.c_int, C.c_char_p, C.c_size_t, C.POINTER(C.c_uint64), C.POINTER(C.c_uint32)]),
    "shodh_index_cfg_default": (None, [C.POINTER(IndexCfg)]),
    "shodh_index_create": (C.c_int, [C.POINTER(IndexCfg), C.POINTER(C.c_void_p)]),
    "shodh_index_destroy": (None, [_vp]),
    "shodh_index_add": (C.c_int, [_vp, _fp, C.c_uint64, C.POINTER(C.c_uint32)]),
    "shodh_index_add_device": (C.c_int, [_vp, _fp, C.c_uint64, C.POINTER(C.c_uint32)]),
    "shodh_index_build": (C.c_int, [_vp, _fp, C.c_uint64]),
    "shodh_index_build_device": (C.c_int, [_vp, _fp, C.c_uint64]),
    "shodh_index_search": (C.c_int, [_vp, _fp, C.c_uint32, C.c_uint32, _u32p, _fp, _u32p]),
    "shodh_index_brute_force_search": (C.c_int, [_vp, _fp, C.c_uint32, C.c_uint32, _u32p, _fp, _u32p]),
    "shodh_index_search_device": (C.c_int, [_vp, _fp, C.c_uint32, C.c_uint32, _u32p, _fp, _u32p, _vp]),
    "shodh_index_set_coalesce": (C.c_int, [_vp, C.c_int, C.c_uint32]),
    "shodh_index_coalesce_stats": (C.c_int, [_vp, C.POINTER(C.c_uint64 * 6), C.c_int]),
    "shodh_index_mark_deleted": (C.c_int, [_vp, C.c_uint32, C.POINTER(C.c_int)]),
    "shodh_index_mark_deleted_batch": (C.c_int, [_vp, _u32p, C.c_uint64, C.POINTER(C.c_uint64)]),
    "shodh_index_is_deleted": (C.c_int, [_vp, C.c_uint32]),
    "shodh_index_len": (C.c_uint64, [_vp]),
    "shodh_index_deleted_count": (C.c_uint64, [_vp]),
    "shodh_index_deletion_ratio": (C.c_float, [_vp]),
    "shodh_index_needs_compaction": (C.c_int, [_vp]),
    "shodh_index_clear_deleted": (C.c_int, [_vp]),
    "shodh_index_extract_rows": (C.c_int, [_vp, C.c_uint64, C.c_uint64, _fp]),
    "shodh_index_extract_live_rows": (C.c_int, [_vp, _fp, _u32p, C.c_uint64, C.POINTER(C.c_uint64)]),
    "shodh_index_dim": (C.c_uint32, [_vp]),
    "shodh_index_stage_timings": (C.c_int, [_vp, C.POINTER(C.c_float * 4)]),
    "shodh_index_kernel_timing": (C.c_int, [_vp, C.c_int, C.POINTER(C.c_float), C.POINTER(C.c_float), C.POINTER(C.c_uint32)]),
    "shodh_index_scan_stats": (C.c_int, [_vp, C.POINTER(C.c_uint64 * 8)]),
    "shodh_index_set_graph": (C.c_int, [_vp, _u32p, _u32p, C.c_uint32, C.c_uint32]),
    "shodh_index_get_graph": (C.c_int, [_vp, _u32p, _u32p, C.c_uint32, C.POINTER(C.c_uint32)]),
    "shodh_index_build_with_graph": (C.c_int, [_vp, _fp, C.c_uint64, _u32p, _u32p, C.c_uint32, C.c_uint32]),
    "shodh_index_incremental_repair": (C.c_int, [_vp, C.c_uint32, C.c_uint32, C.POINTER(C.c_uint32)]),
    "shodh_index_vamana_build": (C.c_int, [_vp, C.c_uint64, _u32p, _u32p, C.c_uint32]),
    "shodh_topk_merge_device": (C.c_int, [_u32p, _fp, C.c_uint32, C.c_uint32, C.c_uint32, _u32p, _fp, _u32p, _vp]),
    "shodh_topk_merge_strided_device": (C.c_int, [_u32p, _fp, C.c_uint64, C.c_uint32, C.c_uint32, C.c_uint32, _u32p, _fp, _u32p, _vp]),
    "shodh_sharded_cfg_default": (None, [C.POINTER(ShardedCfg)]),
    "shodh_sharded_index_create": (C.c_int, [C.POINTER(ShardedCfg), _i32p, C.c_uint32, C.POINTER(C.c_void_p)]),
    "shodh_sharded_index_destroy": (None, [_vp]),
    "shodh_sharded_index_shards": (C.c_uint32, [_vp]),
    "shodh_sharded_index_uses_rccl": (C.c_int, [_vp]),
    "shodh_sharded_index_len": (C.c_uint64, [_vp]),
    "shodh_sharded_index_shard_len": (C.c_uint64, [_vp, C.c_uint32]),
    "shodh_sharded_index_build": (C.c_int, [_vp, _fp, C.c_uint64]),
    "shodh_sharded_index_add": (C.c_int, [_vp, _fp, C.c_uint64, C.POINTER(C.c_uint32)]),
    "shodh_sharded_index_search": (C.c_int, [_vp, _fp, C.c_uint32, C.c_uint32, _u32p, _fp, _u32p]),
    "shodh_sharded_index_search_device": (C.c_int, [_vp, _fp, C.c_uint32, C.c_uint32, _u32p, _fp, _u32p, _vp]),
    "shodh_sharded_index_mark_deleted": (C.c_int, [_vp, C.c_uint32, C.POINTER(C.c_int)]),
    "shodh_sharded_index_mark_deleted_batch": (C.c_int, [_vp, _u32p, C.c_uint64, C.POINTER(C.c_uint64)]),
    "shodh_sharded_index_is_deleted": (C.c_int, [_vp, C.c_uint32]),
    "shodh_sharded_index_deleted_count": (C.c_uint64, [_vp]),
    "shodh_sharded_index_clear_deleted": (C.c_int, [_vp]),
    "shodh_sharded_index_extract_rows": (C.c_int, [_vp, C.c_uint64, C.c_uint64, _fp]),
    "shodh_sharded_index_set_ivfpq": (C.c_int, [_vp, _fp, C.c_uint32, _fp, C.c_uint32, C.c_uint32, _u64p, _u32p, _u8p]),
    "shodh_sharded_index_ivfpq_insert": (C.c_int, [_vp, C.c_uint32, _fp]),
    "shodh_sharded_index_set_coalesce": (C.c_int, [_vp, C.c_int, C.c_uint32]),
    "shodh_sharded_index_coalesce_stats": (C.c_int, [_vp, C.POINTER(C.c_uint64 * 6), C.c_int]),
    "shodh_sharded_index_host_timings": (C.c_int, [_vp, C.POINTER(C.c_float * 4)]),
    "shodh_rccl_info": (C.c_int, [C.c_char_p, C.c_size_t]),
    "shodh_index_set_ivfpq": (C.c_int, [_vp, _fp, C.c_uint32, _fp, C.c_uint32, C.c_uint32, _u64p, _u32p, _u8p]),
    "shodh_index_ivfpq_insert": (C.c_int, [_vp, C.c_uint32, _fp]),
    "shodh_index_ivfpq_encode": (C.c_int, [_vp, _fp, C.c_uint64, _u32p, _u8p]),
    "shodh_ivfpq_train": (C.c_int, [C.c_int, _fp, C.c_uint64, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32, _u32p, _u32p, _fp, _fp]),
    "shodh_cosine_similarity_batch": (C.c_int, [C.c_int, _fp, _fp, C.c_uint64, C.c_uint32, C.c_uint32, _fp]),
    "shodh_top_k_similar": (C.c_int, [C.c_int, _fp, C.c_uint32, _fp, C.c_uint64, C.c_uint32, C.c_uint64, C.c_uint32, _fp, _u32p, C.POINTER(C.c_uint64)]),
    "shodh_embed_cfg_default": (None, [C.POINTER(EmbedCfg)]),
    "shodh_embedder_create": (C.c_int, [C.POINTER(EmbedCfg), C.POINTER(C.c_void_p)]),
    "shodh_embedder_destroy": (None, [_vp]),
    "shodh_embedder_param_count": (C.c_uint64, [_vp]),
    "shodh_embedder_load_weights": (C.c_int, [_vp, _fp, C.c_uint64]),
    "shodh_embedder_init_synthetic": (C.c_int, [_vp, C.c_uint64, _fp, C.c_uint64]),
    "shodh_embedder_load_file": (C.c_int, [_vp, C.c_char_p]),
    "shodh_embedder_load_tensor": (C.c_int, [_vp, C.c_char_p, _fp, C.c_uint64, C.c_uint32]),
    "shodh_embedder_load_quantized": (C.c_int, [_vp, C.c_char_p, _vp, C.c_uint32, C.c_uint32, _fp, _vp, C.c_uint32]),
    "shodh_embedder_finish_weights": (C.c_int, [_vp]),
    "shodh_embedder_weight_source": (C.c_int, [_vp, C.c_char_p, C.POINTER(C.c_uint32)]),
    "shodh_weight_file_open": (C.c_int, [C.c_char_p, C.POINTER(EmbedCfg), C.POINTER(C.c_void_p)]),
    "shodh_weight_file_close": (None, [_vp]),
    "shodh_weight_file_blob": (C.c_int, [_vp, _fp, C.c_uint64]),
    "shodh_weight_file_quantized": (C.c_int, [_vp, C.c_char_p, _vp, C.c_uint64, _fp, _i32p, C.c_uint32, C.POINTER(C.c_uint32)]),
    "shodh_int8_dense_quantized": (C.c_int, [C.c_int, _fp, _vp, C.c_uint32, _fp, _vp, C.c_uint32, _fp, C.c_uint32, C.c_uint32, C.c_uint32, _fp, _i32p,
                                             C.POINTER(C.c_float), C.POINTER(C.c_int32)]),
    "shodh_embed_param_count": (C.c_uint64, [C.POINTER(EmbedCfg)]),
    "shodh_embedder_synthetic_weights": (C.c_int, [C.POINTER(EmbedCfg), C.c_uint64, _fp, C.c_uint64]),
    "shodh_hash_embed": (C.c_int, [C.c_char_p, C.c_size_t, C.c_uint32, _fp]),
    "shodh_finalize_pooled": (C.c_size_t, [_fp, C.c_size_t, C.c_int, C.c_size_t, _fp]),
    "shodh_search_ids_postprocess": (C.c_size_t, [_u32p, _fp, C.c_size_t, _u8p, C.c_size_t, C.c_size_t, _u8p, _fp]),
    "shodh_vama_info_read": (C.c_int, [C.c_char_p, C.POINTER(VamaInfo)]),
    "shodh_vama_load": (C.c_int, [C.c_char_p, _fp, _u32p, C.c_void_p, _u32p]),
    "shodh_vama_save": (C.c_int, [C.c_char_p, _fp, C.c_uint64, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint8, _u32p, C.c_uint32, C.c_uint64, C.c_void_p, _u32p]),
    "shodh_span_info_read": (C.c_int, [C.c_char_p, C.POINTER(SpanInfo)]),
    "shodh_span_load": (C.c_int, [C.c_char_p, _fp, _fp, _u64p, _u32p, _u8p]),
    "shodh_span_save": (C.c_int, [C.c_char_p, C.c_uint64, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint8, _fp, _fp, _u64p, _u32p, _u8p]),
    "shodh_rrf_fuse": (C.c_size_t, [C.c_float, _fp, C.c_size_t, _u8p, C.POINTER(C.c_size_t), _u8p, _fp, C.c_size_t]),
    "shodh_calculate_tag_score": (C.c_float, [C.c_char_p, C.POINTER(C.c_char_p), C.c_size_t]),
    "shodh_apply_recency_boost": (C.c_float, [C.c_float, C.c_int64, C.c_uint64, C.c_float]),
    "shodh_relevance_cfg_default": (None, [C.POINTER(RelevanceCfg)]),
    "shodh_rank_surfaced": (C.c_size_t, [C.POINTER(Weights), C.POINTER(RelevanceCfg), C.c_size_t] + [_fp] * 5 + [_u32p, _fp, _vp, _vp, _u8p, _u32p, _fp, _u8p]),
    "shodh_leg_fusion_cfg_default": (None, [C.POINTER(LegFusionCfg)]),
    "shodh_density_weights": (None, [C.c_float, _fp]),
    "shodh_leg_fusion_weights": (None, [C.c_int, C.c_float, C.c_float, C.c_float, C.POINTER(C.c_float), C.POINTER(C.c_float)]),
    "shodh_fuse_legs": (C.c_size_t, [C.POINTER(LegFusionCfg), _u8p, _fp, _fp, C.c_size_t, _u8p, _fp, C.c_size_t, C.c_size_t, _u8p, _fp,
                                     C.c_size_t, C.POINTER(C.c_float)]),
    "shodh_embedder_dimension": (C.c_uint32, [_vp]),
    "shodh_embedder_encode_ids": (C.c_int, [_vp, _i32p, _u8p, C.c_uint32, _fp]),
    "shodh_embedder_encode_ids_device": (C.c_int, [_vp, _i32p, _u8p, C.c_uint32, _fp, _vp]),
    "shodh_int8_dense": (C.c_int, [C.c_int, _fp, _fp, _fp, C.c_uint32, C.c_uint32, C.c_uint32, _fp, _i32p, C.POINTER(C.c_float), C.POINTER(C.c_int32), C.POINTER(C.c_float)]),
    "shodh_embedder_stage_timings": (C.c_int, [_vp, C.POINTER(C.c_float * 2)]),
    "shodh_embedder_encode_ids_scoped": (C.c_int, [_vp, _i32p, _u8p, C.c_uint32, C.c_uint32, _fp]),
    "shodh_embedder_encode_ids_device_scoped": (C.c_int, [_vp, _i32p, _u8p, C.c_uint32, C.c_uint32, _fp, _vp]),
    "shodh_embedder_set_coalesce": (C.c_int, [_vp, C.c_int, C.c_uint32]),
    "shodh_embedder_coalesce_stats": (C.c_int, [_vp, C.POINTER(C.c_uint64 * 6), C.c_int]),
    "shodh_embedder_set_quant_scope": (C.c_int, [_vp, C.c_uint32]),
    "shodh_embedder_quant_scope": (C.c_uint32, [_vp]),
    "shodh_weights_default": (None, [C.POINTER(Weights)]),
    "shodh_weights_normalize": (None, [C.POINTER(Weights)]),
    "shodh_weights_apply_feedback": (None, [C.POINTER(Weights), C.c_int, C.c_int, C.c_int, C.c_int]),
    "shodh_calibrate_score": (C.c_float, [C.c_float]),
    "shodh_fuse_scores": (C.c_float, [C.POINTER(Weights)] + [C.c_float] * 4),
    "shodh_fuse_scores_with_momentum": (C.c_float, [C.POINTER(Weights)] + [C.c_float] * 5),
    "shodh_fuse_scores_full": (C.c_float, [C.POINTER(Weights)] + [C.c_float] * 5 + [C.c_uint32, C.c_float]),
    "shodh_fuse_scores_full_batch": (C.c_int, [C.c_int, C.POINTER(Weights), C.c_uint64] + [_fp] * 5 + [_u32p, _fp, _fp]),
}

_lib = None


def lib():
    """Loads libshodh_hip.so. Raises if it is missing (build it with `python -m shodh_memory_amd.build`)."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise ImportError("libshodh_hip.so is not built (%s). Run `python -m shodh_memory_amd.build`; "
                          "there is no CPU fallback." % LIB_PATH)
    L = C.CDLL(LIB_PATH)
    for name, (res, args) in SYMBOLS.items():
        f = getattr(L, name)      # AttributeError if the library does not export a declared symbol
        f.restype = res
        f.argtypes = args
    _lib = L
    return L


def last_error():
    return lib().shodh_last_error().decode("utf-8", "replace")


def check(rc):
    if rc != OK:
        raise ShodhError(rc, last_error())
    return rc
