"""Hot subset of `RetrievalEngine` (src/memory/retrieval.rs): IdMapping, index_memory, search_ids,
search_by_embedding -- the glue between the embedder, the vector index and memory ids (the structural chunker
is mirrored in chunking.py). The storage engine (RocksDB) and event buffer of the reference are out of scope; an in-memory mapping
stands in for `update_vector_mapping`.
"""
import hashlib
import uuid as _uuid
from collections import OrderedDict

import numpy as np

from . import _lib as L
from .index import VamanaConfig, VamanaIndex

VECTOR_SEARCH_CANDIDATE_MULTIPLIER = 2      # src/constants.rs:350
POLAR_QUERY_VECTOR_POOL_MULTIPLIER = 2      # src/constants.rs:453
EMBEDDING_CACHE_CAPACITY = 2_000            # memory/mod.rs:800-801 (moka caches; a plain LRU here)


class IdMapping:
    """retrieval.rs:70-142: vector_id (u32) <-> MemoryId (uuid); several chunk vectors may map to one memory.
    `insert` / `insert_chunks` REPLACE whatever the memory was mapped to before (re-indexing must not leave orphans)."""

    def __init__(self):
        self.vector_to_memory = {}
        self.memory_to_vectors = {}

    def _drop(self, memory_id):
        for old in self.memory_to_vectors.pop(memory_id, []):
            self.vector_to_memory.pop(old, None)

    def insert(self, memory_id, vector_id):
        self._drop(memory_id)                                     # retrieval.rs:89-99
        self.vector_to_memory[vector_id] = memory_id
        self.memory_to_vectors[memory_id] = [vector_id]

    def insert_chunks(self, memory_id, vector_ids):
        self._drop(memory_id)                                     # retrieval.rs:105-117
        for v in vector_ids:
            self.vector_to_memory[v] = memory_id
        self.memory_to_vectors[memory_id] = list(vector_ids)

    def get_memory_id(self, vector_id):
        return self.vector_to_memory.get(vector_id)

    def get_vector_ids(self, memory_id):
        return list(self.memory_to_vectors.get(memory_id, []))

    def remove_all(self, memory_id):
        """-> ALL vector ids the memory had (retrieval.rs:124-133)"""
        removed = self.memory_to_vectors.pop(memory_id, [])
        for v in removed:
            self.vector_to_memory.pop(v, None)
        return removed

    def len(self):
        return len(self.memory_to_vectors)

    def clear(self):
        self.memory_to_vectors.clear()
        self.vector_to_memory.clear()


class RetrievalEngine:
    def __init__(self, embedder, dimension=384, device=0, scan_mode=L.SCAN_AUTO):
        # retrieval.rs:174-193: NormalizedDotProduct only
        self.embedder = embedder
        self.vector_index = VamanaIndex(VamanaConfig(dimension=dimension, device=device, scan_mode=scan_mode))
        self.id_mapping = IdMapping()

    def index_memory(self, memory_id, content=None, embedding=None, chunks=None):
        """retrieval.rs:646-730: content longer than the embedder's token window is split by the structural chunker
        (budget = `chunk_budget_tokens()`, counted with `count_tokens`) and every chunk gets its own vector, all mapped to the
        memory; short content is one vector, a pre-computed `embedding` reused (:704-705). `chunks` passes pre-chunked texts
        straight in (tests)."""
        from .chunking import ChunkConfig, chunk_text
        assert isinstance(memory_id, _uuid.UUID)
        if chunks is None and content is not None:
            r = chunk_text(content, ChunkConfig.for_budget(self.embedder.chunk_budget_tokens()), self.embedder.count_tokens)
            if r.was_chunked:
                chunks = r.chunks
        if chunks:
            # the reference encodes the chunks one by one (:668-676, one encode() each): N x encode() in one device call. INT8: every
            # DynamicQuantizeLinear range spans ONE chunk, so the vectors do not depend on the other chunks (encode_batch's would)
            vecs = self.embedder.encode_each(chunks)
        else:
            vecs = [np.asarray(embedding, np.float32) if embedding is not None else self.embedder.encode(content)]
        ids = [self.vector_index.add_vector(v) for v in vecs]
        if chunks:
            self.id_mapping.insert_chunks(memory_id, ids)        # :690-693
        else:
            self.id_mapping.insert(memory_id, ids[0])             # :715
        return ids

    def _postprocess(self, results, limit, exclude=None):
        """retrieval.rs:920-963. Only the returned vector ids are mapped (O(k) per query, like the reference's hash-map
        lookups): the C entry point takes a vector_to_memory TABLE indexed by vector id, so the results are renumbered
        0..m-1 in their rank order -- the per-memory max over chunks walks them in that order and the final sort is by
        (similarity, MemoryId), neither depends on the vector id itself."""
        if not results:
            return []
        m_in = len(results)
        v2m = np.full((m_in, 16), 0xFF, np.uint8)
        for i, (vid, _) in enumerate(results):
            mid = self.id_mapping.get_memory_id(vid)
            if mid is not None and (exclude is None or mid != exclude):
                v2m[i] = np.frombuffer(mid.bytes, np.uint8)
        vec_ids = np.arange(m_in, dtype=np.uint32)
        dists = np.array([r[1] for r in results], np.float32)
        out_u = np.zeros((max(limit, 1), 16), np.uint8)
        out_s = np.zeros(max(limit, 1), np.float32)
        m = L.lib().shodh_search_ids_postprocess(vec_ids.ctypes.data, dists.ctypes.data, m_in, v2m.ctypes.data, m_in, limit,
                                                 out_u.ctypes.data, out_s.ctypes.data)
        return [(_uuid.UUID(bytes=bytes(out_u[i])), float(out_s[i])) for i in range(m)]

    def search_ids(self, query_text=None, query_embedding=None, limit=10):
        """retrieval.rs:872-964 -> [(MemoryId, similarity)]"""
        if query_embedding is None:
            if query_text is None:
                return []
            query_embedding = self.embedder.encode(query_text)
        res = self.vector_index.search(query_embedding, limit * VECTOR_SEARCH_CANDIDATE_MULTIPLIER * 2)   # :913-918
        return self._postprocess(res, limit)

    def search_by_embedding(self, embedding, limit, exclude_id=None):
        """retrieval.rs:980-1031 (dedup / interference check of remember, memory/mod.rs:1252-1256)"""
        res = self.vector_index.search(embedding, limit * VECTOR_SEARCH_CANDIDATE_MULTIPLIER * 2)
        return self._postprocess(res, limit, exclude=exclude_id)

    # -- index maintenance (retrieval.rs:1556-1627, :1658-1670) ---------------------------------------------
    def index_health(self):
        """IndexHealth (retrieval.rs:1658-1684)"""
        from .index import DELETION_RATIO_THRESHOLD, REBUILD_THRESHOLD
        ix = self.vector_index
        return dict(total_vectors=ix.len(), incremental_inserts=ix.incremental_insert_count(), deleted_count=ix.deleted_count(),
                    deletion_ratio=ix.deletion_ratio(), needs_rebuild=ix.needs_rebuild(), needs_compaction=ix.needs_compaction(),
                    rebuild_threshold=REBUILD_THRESHOLD, deletion_ratio_threshold=DELETION_RATIO_THRESHOLD)

    def force_quality_rebuild(self):
        """retrieval.rs:1556-1627: rebuild the index from the MAPPED vectors in vector-id order (unmapped = soft-deleted
        vectors drop out), new ids are positional, the mapping is regrouped per memory. For the exact index the rebuild
        changes no result; what it does is compact and reset the incremental counter, like the reference's."""
        ix = self.vector_index
        if ix.is_empty() or ix.incremental_insert_count() == 0:
            return
        all_vectors = ix.extract_all_vectors()
        pairs = sorted(self.id_mapping.vector_to_memory.items())
        if any(vid >= len(all_vectors) for vid, _ in pairs):
            raise L.ShodhError(L.ERR_STATE, "quality rebuild aborted: %d mapped vectors but %d extractable"
                               % (len(pairs), sum(1 for vid, _ in pairs if vid < len(all_vectors))))
        ix.rebuild_from_vectors(np.stack([all_vectors[vid] for vid, _ in pairs]) if pairs else np.zeros((0, ix.config.dimension), np.float32))
        per_memory = {}
        for new_id, (_, mid) in enumerate(pairs):
            per_memory.setdefault(mid, []).append(new_id)
        self.id_mapping.clear()
        for mid, vids in per_memory.items():
            self.id_mapping.insert_chunks(mid, vids)


class _EmbeddingCache:
    """SHA-256(text) -> embedding, bounded (memory/mod.rs:800-801, :5856-5860)."""

    def __init__(self, capacity=EMBEDDING_CACHE_CAPACITY):
        self.capacity, self.d, self.hits, self.misses = capacity, OrderedDict(), 0, 0

    def get_or(self, text, encode):
        key = hashlib.sha256(text.encode("utf-8")).digest()
        if key in self.d:
            self.hits += 1
            self.d.move_to_end(key)
            return self.d[key]
        self.misses += 1
        v = np.asarray(encode(text), np.float32)
        self.d[key] = v
        if len(self.d) > self.capacity:
            self.d.popitem(last=False)
        return v


class MemoryPathSlice:
    """The embed-and-index slice of `MemorySystem::remember` (memory/mod.rs:1025-1051, :1088-1094, :1252-1256) and the
    query-embed + vector leg of `MemorySystem::recall` (:2873-2904, :3580-3617) over a `RetrievalEngine`. Everything else in
    those two 1000-line functions (storage, graph, BM25, temporal facts, ...) is the reference's host code and out of scope."""

    def __init__(self, retriever):
        self.retriever = retriever
        self.embedder = retriever.embedder
        self.content_cache = _EmbeddingCache()
        self.query_cache = _EmbeddingCache()

    def remember(self, memory_id, content, embeddings=None):
        """-> (embedding, indexed, similar): cache-or-encode the content, index it at once, then the k = 5 similarity search
        that feeds the interference check, excluding the new memory itself."""
        if embeddings is None:
            embeddings = self.content_cache.get_or(content, self.embedder.encode)
        indexed = True
        try:
            self.retriever.index_memory(memory_id, content=content, embedding=embeddings)
        except L.ShodhError:
            indexed = False                                         # the reference logs and carries on (:1088-1094)
        similar = self.retriever.search_by_embedding(embeddings, 5, exclude_id=memory_id)
        return embeddings, indexed, similar

    def recall_vector_leg(self, query_text, max_results, query_embedding=None, polarity_sensitive=False, negated_embedding=None,
                          episode_candidates=None):
        """-> [(MemoryId, similarity)]: a pre-computed embedding wins, else the SHA-256-keyed query cache, else encode_query;
        vector_top_k = max_results * 3 (* 2 for polarity-sensitive queries); with a negated-form embedding the two result lists
        are united per memory (best score), ordered (score desc, id asc); an episode filter keeps its candidates only."""
        if query_embedding is None or len(query_embedding) == 0:
            query_embedding = self.query_cache.get_or(query_text, self.embedder.encode_query)
        polar_mul = max(POLAR_QUERY_VECTOR_POOL_MULTIPLIER, 1) if polarity_sensitive else 1
        vector_top_k = max_results * 3 * polar_mul
        vr = self.retriever.search_ids(query_embedding=query_embedding, limit=vector_top_k)
        if negated_embedding is not None:
            best = {}
            for mid, score in vr + self.retriever.search_ids(query_embedding=negated_embedding, limit=vector_top_k):
                if mid not in best or score > best[mid]:
                    best[mid] = score
            key = lambda t: (-_total_order(t[1]), t[0].bytes)
            vr = sorted(best.items(), key=key)
        if episode_candidates is not None:
            vr = [(mid, s) for mid, s in vr if mid in episode_candidates]
        return vr


def _total_order(x):
    """f32::total_cmp as an integer key"""
    b = int(np.array([x], np.float32).view(np.int32)[0])
    return b ^ (((b >> 31) & 0xFFFFFFFF) >> 1)
