"""shodh_memory_amd -- MI355X-native embed-and-recall hot path of shodh-memory.

Everything here sits on top of libshodh_hip.so (hand-written HIP for gfx950, C ABI in
include/shodh_hip.h). Importing this package loads that library and fails loudly if it is
missing: there is no CPU fallback and the oracle under oracle/ is never used by the product.
"""
import sys as _sys

from . import _lib
from ._lib import ShodhError, lib

# Fail loudly at import time if the HIP library is not built or is stale. The one exception is
# `python -m shodh_memory_amd.build` (the parent package is imported with argv[0] == "-m" before
# the build module runs): there the library is about to be (re)built and is loaded lazily.
if not (_sys.argv and _sys.argv[0] == "-m"):
    lib()

from .index import (BackendConfig, BackendType, DistanceMetric, SpannIndex, VamanaConfig,   # noqa: E402
                    VamanaIndex, VectorIndexBackend)
from .relevance import (LearnedWeights, LegFusion, apply_recency_boost, calculate_density_weights, calculate_tag_score,  # noqa: E402
                        calibrate_score, rank_surfaced)                                      # noqa: E402
from .similarity import cosine_similarity, top_k_similar                                # noqa: E402
from .embedder import Embedder, MiniLMEmbedder                                              # noqa: E402
from .retrieval import IdMapping, MemoryPathSlice, RetrievalEngine                                           # noqa: E402

__all__ = ["ShodhError", "lib", "VamanaIndex", "VamanaConfig", "VectorIndexBackend", "BackendConfig", "BackendType",
           "DistanceMetric", "SpannIndex", "LearnedWeights", "calibrate_score", "Embedder", "MiniLMEmbedder",
           "IdMapping", "RetrievalEngine", "MemoryPathSlice", "LegFusion", "calculate_density_weights", "calculate_tag_score",
           "apply_recency_boost", "rank_surfaced", "cosine_similarity", "top_k_similar"]
