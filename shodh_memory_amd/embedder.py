"""Host-side mirror of the reference's `Embedder` trait and `MiniLMEmbedder`
(src/embeddings/mod.rs:52-88, src/embeddings/minilm.rs) over the C ABI.

Tokenisation (HF `tokenizers`, WordPiece, truncation pinned to 128: minilm.rs:112-117) stays on the
host exactly as in the reference; the device takes token ids. No tokenizer.json / weights ship with
the reference tree (downloaded at first run, embeddings/downloader.rs:29-53), so the caller supplies
them: a `tokenizers.Tokenizer` and either a weight blob or a synthetic seed.
"""
import ctypes as C
import os

import numpy as np

from . import _lib as L

MODEL_TOKEN_WINDOW = 128        # embeddings/chunking.rs:60
SPECIAL_TOKEN_OVERHEAD = 2      # embeddings/chunking.rs:63  ([CLS] + [SEP])


def embed_cfg(**kw):
    cfg = L.EmbedCfg()
    L.lib().shodh_embed_cfg_default(C.byref(cfg))
    for k, v in kw.items():
        setattr(cfg, k, v)
    return cfg


def param_count(cfg=None):
    cfg = cfg or embed_cfg()
    return int(L.lib().shodh_embed_param_count(C.byref(cfg)))


def synthetic_weights(seed, cfg=None):
    """The deterministic synthetic parameter blob (HF BertModel order). Host only."""
    cfg = cfg or embed_cfg()
    n = param_count(cfg)
    blob = np.empty(n, np.float32)
    L.check(L.lib().shodh_embedder_synthetic_weights(C.byref(cfg), int(seed), blob.ctypes.data, n))
    return blob


def blob_to_state_dict(blob, cfg=None):
    """Splits a blob into HF `BertModel` parameter names -> arrays (for checkers / weight import)."""
    cfg = cfg or embed_cfg()
    H, I = cfg.hidden, cfg.intermediate
    out, o = {}, 0

    def take(name, *shape):
        nonlocal o
        n = int(np.prod(shape))
        out[name] = blob[o:o + n].reshape(shape)
        o += n
    take("embeddings.word_embeddings.weight", cfg.vocab, H)
    take("embeddings.position_embeddings.weight", cfg.max_pos, H)
    take("embeddings.token_type_embeddings.weight", cfg.type_vocab, H)
    take("embeddings.LayerNorm.weight", H); take("embeddings.LayerNorm.bias", H)
    for l in range(cfg.layers):
        p = "encoder.layer.%d." % l
        for nm in ("query", "key", "value"):
            take(p + "attention.self.%s.weight" % nm, H, H); take(p + "attention.self.%s.bias" % nm, H)
        take(p + "attention.output.dense.weight", H, H); take(p + "attention.output.dense.bias", H)
        take(p + "attention.output.LayerNorm.weight", H); take(p + "attention.output.LayerNorm.bias", H)
        take(p + "intermediate.dense.weight", I, H); take(p + "intermediate.dense.bias", I)
        take(p + "output.dense.weight", H, I); take(p + "output.dense.bias", H)
        take(p + "output.LayerNorm.weight", H); take(p + "output.LayerNorm.bias", H)
    assert o == blob.size
    return out


def state_dict_to_blob(sd, cfg=None):
    """Inverse of blob_to_state_dict: accepts HF `BertModel` tensors/arrays (e.g. from model.safetensors)."""
    cfg = cfg or embed_cfg()
    ref = blob_to_state_dict(np.zeros(param_count(cfg), np.float32), cfg)
    parts = []
    for name, z in ref.items():
        a = sd[name]
        a = a.detach().cpu().numpy() if hasattr(a, "detach") else np.asarray(a)
        assert a.shape == z.shape, (name, a.shape, z.shape)
        parts.append(np.ascontiguousarray(a, np.float32).reshape(-1))
    return np.concatenate(parts)


def finalize_pooled(pooled, apply_prenorm=False, dimension=None):
    """MiniLMEmbedder::finalize_pooled (minilm.rs:846-878): scrub, optional parameter-free LayerNorm (nomic), truncate, L2."""
    p = np.ascontiguousarray(pooled, np.float32).reshape(-1)
    out = np.zeros(max(p.size, 1), np.float32)
    m = L.lib().shodh_finalize_pooled(p.ctypes.data, p.size, int(bool(apply_prenorm)), p.size if dimension is None else int(dimension), out.ctypes.data)
    return out[:m].copy()


class WeightFile:
    """Host-only view of a model file (shodh_weight_file_*): what shodh_embedder_load_file parses, without a device."""

    def __init__(self, path, cfg=None):
        self._cfg = cfg or embed_cfg()
        self._h = C.c_void_p()
        L.check(L.lib().shodh_weight_file_open(os.fsencode(path), C.byref(self._cfg), C.byref(self._h)))

    def close(self):
        if self._h and self._h.value:
            L.lib().shodh_weight_file_close(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def blob(self):
        n = param_count(self._cfg)
        b = np.empty(n, np.float32)
        L.check(L.lib().shodh_weight_file_blob(self._h, b.ctypes.data, n))
        return b

    def quantized(self, name, shape):
        """-> None if the file stores `name` as floats, else (q int8 [N, K] signed storage, scale [n], zero_point int32 [n] in the same signed terms)"""
        n = C.c_uint32()
        L.check(L.lib().shodh_weight_file_quantized(self._h, name.encode(), None, 0, None, None, 0, C.byref(n)))
        if n.value == 0:
            return None
        q = np.empty(shape, np.int8); sc = np.empty(n.value, np.float32); zp = np.empty(n.value, np.int32)
        L.check(L.lib().shodh_weight_file_quantized(self._h, name.encode(), q.ctypes.data, q.size, sc.ctypes.data, zp.ctypes.data, n.value, C.byref(n)))
        return q, sc, zp


def estimate_tokens(text):
    """token_estimation.rs:36-88: content-aware estimate used by `count_tokens` when no tokenizer is loaded. A 512-byte
    sample decides the mode: CJK (more than 2.5 bytes per char) -> ceil(1.5 * chars); code (>= 8 % syntax punctuation in the
    sample) -> bytes * 10 / 32; prose -> ceil(bytes / 4)."""
    if text == "":
        return 0
    b = text.encode("utf-8")
    byte_len = len(b)
    sample = b[:min(byte_len, 512)]
    syntax = sum(1 for x in sample if x in b'{}[]();=<>|&#@!~^\\"\'')
    high = sum(1 for x in sample if x >= 0xC0)
    if high > 0:
        chars = len(text)
        if chars > 0 and byte_len > chars * 2 + chars // 2:
            return (chars * 3 + 1) // 2
    if len(sample) > 0 and syntax * 100 >= len(sample) * 8:
        return byte_len * 10 // 32
    return (byte_len + 3) // 4


class Embedder:
    """trait Embedder (embeddings/mod.rs:52-88)"""

    def encode(self, text):
        raise NotImplementedError

    def encode_query(self, text):
        return self.encode(text)

    def dimension(self):
        raise NotImplementedError

    def encode_batch(self, texts):
        return [self.encode(t) for t in texts]

    def count_tokens(self, text):
        return estimate_tokens(text) + SPECIAL_TOKEN_OVERHEAD

    def chunk_budget_tokens(self):
        return MODEL_TOKEN_WINDOW


class MiniLMEmbedder(Embedder):
    def __init__(self, tokenizer=None, weights=None, synthetic_seed=None, dtype=L.DTYPE_BF16, device=0,
                 query_prefix="", doc_prefix="", max_length=256, simplified=False, dim=384, compute_padded=None, weights_path=None,
                 quant_scope=L.QUANT_SCOPE_BATCH, **cfg_kw):
        self._dim = dim
        self.simplified_mode = simplified
        self.query_prefix, self.doc_prefix = query_prefix, doc_prefix
        self.max_length = max_length
        self.tokenizer = tokenizer
        self._h = C.c_void_p()
        if simplified:
            return
        if tokenizer is not None:
            # minilm.rs:112-117: truncation pinned to the model window; padding is done by the embedder
            tokenizer.enable_truncation(max_length=MODEL_TOKEN_WINDOW)
            tokenizer.no_padding()
        if compute_padded is None:
            compute_padded = dtype == L.DTYPE_INT8      # the reference's INT8 tensor is padded to max_length (minilm.rs:588-593)
        # weights_path: the model file itself, as EmbeddingConfig.model_path names it (minilm.rs:212-220): model.safetensors, model.onnx or the
        # dynamic-quantisation export -- with dtype INT8 the export's own 8-bit tensors, scales and zero points are what the device multiplies
        # quant_scope (INT8 only): the default scope of encode_ids. BATCH = the reference's encode_batch (ranges over the batch tensor, minilm.rs:996-1115);
        # PER_TEXT = N x encode() (ranges per text, minilm.rs:883-982: what remember / recall compute). See encode_batch / encode_each below.
        cfg = embed_cfg(device=device, dtype=dtype, max_len=max_length, hidden=dim, compute_padded=int(bool(compute_padded)), quant_scope=int(quant_scope),
                        weights_path=os.fsencode(weights_path) if weights_path is not None else None, **cfg_kw)      # cfg_kw: layers=, vocab=, ... (other BERT sizes)
        L.check(L.lib().shodh_embedder_create(C.byref(cfg), C.byref(self._h)))
        cfg.weights_path = None
        if weights is not None:
            w = np.ascontiguousarray(weights, np.float32).reshape(-1)
            L.check(L.lib().shodh_embedder_load_weights(self._h, w.ctypes.data, w.size))
        elif synthetic_seed is not None:
            L.check(L.lib().shodh_embedder_init_synthetic(self._h, int(synthetic_seed), None, 0))
        self._cfg = cfg

    @classmethod
    def new_simplified(cls, dim=384):
        """MiniLMEmbedder::new_simplified (minilm.rs:721-747): hash embeddings, no model."""
        return cls(simplified=True, dim=dim)

    def close(self):
        if self._h and self._h.value:
            L.lib().shodh_embedder_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def dimension(self):
        return self._dim

    # -- weights as files / tensors hand them over (shodh_embedder_load_file and friends) ---------
    def load_file(self, path):
        L.check(L.lib().shodh_embedder_load_file(self._h, os.fsencode(path)))

    def load_tensor(self, name, data, transposed=False):
        a = np.ascontiguousarray(data, np.float32)
        L.check(L.lib().shodh_embedder_load_tensor(self._h, name.encode(), a.ctypes.data, a.size, int(bool(transposed))))

    def load_quantized(self, name, q, scale, zero_point=None, transposed=False):
        """q uint8 or int8, [N, K] (or [K, N] with transposed=True, as ONNX MatMul constants are stored); scale / zero_point scalars or [N]"""
        q = np.ascontiguousarray(q)
        assert q.dtype in (np.uint8, np.int8)
        sc = np.ascontiguousarray(np.atleast_1d(scale), np.float32)
        zp = None if zero_point is None else np.ascontiguousarray(np.atleast_1d(zero_point), q.dtype)
        assert zp is None or zp.size == sc.size
        L.check(L.lib().shodh_embedder_load_quantized(self._h, name.encode(), q.ctypes.data, int(q.dtype == np.int8), int(bool(transposed)), sc.ctypes.data,
                                                      zp.ctypes.data if zp is not None else None, sc.size))

    def finish_weights(self):
        L.check(L.lib().shodh_embedder_finish_weights(self._h))

    def weight_source(self, name):
        out = C.c_uint32()
        L.check(L.lib().shodh_embedder_weight_source(self._h, name.encode(), C.byref(out)))
        return int(out.value)

    # -- device entry points -------------------------------------------------------------------
    def set_quant_scope(self, scope):
        """INT8: which tensor a DynamicQuantizeLinear range spans when a call carries several texts (shodh_embedder_set_quant_scope)."""
        L.check(L.lib().shodh_embedder_set_quant_scope(self._h, int(scope)))

    def quant_scope(self):
        return int(L.lib().shodh_embedder_quant_scope(self._h))

    def encode_ids(self, ids, mask, scope=None):
        """ids int32 [b, max_len], mask uint8 [b, max_len] -> float32 [b, dim] unit rows (zeros if mask empty).
        scope (INT8 only): L.QUANT_SCOPE_BATCH / L.QUANT_SCOPE_PER_TEXT for this call; None = the handle's setting."""
        ids = np.ascontiguousarray(ids, np.int32).reshape(-1, self.max_length)
        mask = np.ascontiguousarray(mask, np.uint8).reshape(-1, self.max_length)
        out = np.zeros((ids.shape[0], self._dim), np.float32)
        if scope is None:
            L.check(L.lib().shodh_embedder_encode_ids(self._h, ids.ctypes.data, mask.ctypes.data, ids.shape[0], out.ctypes.data))
        else:       # the scope travels with the call (no set / restore on the handle: concurrent callers cannot disturb each other)
            L.check(L.lib().shodh_embedder_encode_ids_scoped(self._h, ids.ctypes.data, mask.ctypes.data, ids.shape[0], int(scope), out.ctypes.data))
        return out

    def encode_ids_device(self, ids, mask, out=None, stream=None, scope=None):
        import torch
        b = ids.shape[0]
        if out is None:
            out = torch.empty((b, self._dim), dtype=torch.float32, device=ids.device)
        st = stream if stream is not None else torch.cuda.current_stream(ids.device).cuda_stream
        if scope is None:
            L.check(L.lib().shodh_embedder_encode_ids_device(self._h, ids.data_ptr(), mask.data_ptr(), b, out.data_ptr(), C.c_void_p(st)))
        else:
            L.check(L.lib().shodh_embedder_encode_ids_device_scoped(self._h, ids.data_ptr(), mask.data_ptr(), b, int(scope), out.data_ptr(), C.c_void_p(st)))
        return out

    def set_coalesce(self, enabled, linger_us=30):
        """concurrent one-text encode calls share one per-text forward (shodh_embedder_set_coalesce; on by default)"""
        L.check(L.lib().shodh_embedder_set_coalesce(self._h, int(bool(enabled)), int(linger_us)))

    def coalesce_stats(self, reset=False):
        a = (C.c_uint64 * 6)()
        L.check(L.lib().shodh_embedder_coalesce_stats(self._h, C.byref(a), int(bool(reset))))
        return dict(passes=int(a[0]), calls=int(a[1]), largest=int(a[2]), lingered=int(a[3]), pass_us=int(a[4]), linger_us=int(a[5]))

    def stage_timings_us(self):
        a = (C.c_float * 2)()
        L.check(L.lib().shodh_embedder_stage_timings(self._h, C.byref(a)))
        return dict(embedding_us=a[0], tokens=int(a[1]))

    # -- trait Embedder --------------------------------------------------------------------------
    def _tokenize(self, texts):
        ids = np.zeros((len(texts), self.max_length), np.int32)
        mask = np.zeros((len(texts), self.max_length), np.uint8)
        for i, enc in enumerate(self.tokenizer.encode_batch(list(texts), add_special_tokens=True)):
            n = min(len(enc.ids), self.max_length)                     # minilm.rs:912-921
            ids[i, :n] = enc.ids[:n]
            mask[i, :n] = enc.attention_mask[:n]
        return ids, mask

    def _hash_embed(self, text):
        b = text.encode("utf-8")
        out = np.zeros(self._dim, np.float32)
        L.check(L.lib().shodh_hash_embed(b, len(b), self._dim, out.ctypes.data))
        return out

    def _encode_prefixed(self, text, prefix):                          # minilm.rs:1122-1192
        if text == "":
            return np.zeros(self._dim, np.float32)
        if self.simplified_mode:
            return self._hash_embed(prefix + text if prefix else text)
        if self.tokenizer is None:
            raise L.ShodhError(L.ERR_STATE, "no tokenizer: pass token ids to encode_ids or construct with a tokenizers.Tokenizer")
        ids, mask = self._tokenize([prefix + text if prefix else text])
        return self.encode_ids(ids, mask)[0]

    def encode(self, text):
        return self._encode_prefixed(text, self.doc_prefix)

    def encode_query(self, text):
        return self._encode_prefixed(text, self.query_prefix)

    def encode_each(self, texts):
        """N x encode() in one device call -- the function `remember` / `index_memory` / `recall` compute, text by text (memory/mod.rs:1037,
        retrieval.rs:673, :708, :878: one session.run on [1, max_len] per text, minilm.rs:883-982). INT8: SHODH_QUANT_SCOPE_PER_TEXT, every
        DynamicQuantizeLinear range spans ONE text's padded tensor, so the result is bit-identical to [encode(t) for t in texts] whatever the
        batch. fp32 / bf16: texts never interact; this call (like every one-text call) runs the kernel forms a single text takes whatever the size of
        the batch, so the bytes equal [encode(t) for t in texts] too (encode_batch may pick faster forms for a big batch: bf16 rounding-level differences)."""
        return self.encode_batch(texts, _scope=L.QUANT_SCOPE_PER_TEXT)

    def encode_batch(self, texts, _scope=L.QUANT_SCOPE_BATCH):        # minilm.rs:1247-1376
        """The reference's Embedder::encode_batch: ONE session.run on [B, max_len] (minilm.rs:996-1115; reached from the entity / fact side
        paths, memory/mod.rs:8443, :8838). INT8: SHODH_QUANT_SCOPE_BATCH -- the DynamicQuantizeLinear ranges span the whole batch tensor, so
        a text's embedding depends (slightly) on its batch mates, exactly as in the reference. For bulk ingest of independent memories
        (N x `remember`) use encode_each."""
        texts = list(texts)
        if not texts:
            return []
        out = [np.zeros(self._dim, np.float32) for _ in texts]
        idx = [i for i, t in enumerate(texts) if t != ""]              # empty positions keep zero vectors (:1319-1350)
        if not idx:
            return out
        full = [(self.doc_prefix + texts[i]) if self.doc_prefix else texts[i] for i in idx]
        if self.simplified_mode:
            for i, t in zip(idx, full):
                out[i] = self._hash_embed(t)
            return out
        if self.tokenizer is None:
            raise L.ShodhError(L.ERR_STATE, "no tokenizer")
        ids, mask = self._tokenize(full)
        emb = self.encode_ids(ids, mask, scope=_scope)
        for j, i in enumerate(idx):
            out[i] = emb[j]
        return out

    def count_tokens(self, text):                                      # minilm.rs:1216-1229
        if text == "":
            return SPECIAL_TOKEN_OVERHEAD
        if not self.simplified_mode and self.tokenizer is not None:
            self.tokenizer.no_truncation()
            try:
                return len(self.tokenizer.encode(text, add_special_tokens=True).ids)
            finally:
                self.tokenizer.enable_truncation(max_length=MODEL_TOKEN_WINDOW)
        return estimate_tokens(text) + SPECIAL_TOKEN_OVERHEAD

    def chunk_budget_tokens(self):                                     # minilm.rs:1236-1245
        if not self.doc_prefix:
            return MODEL_TOKEN_WINDOW
        prefix_tokens = max(0, self.count_tokens(self.doc_prefix) - SPECIAL_TOKEN_OVERHEAD)
        return max(0, MODEL_TOKEN_WINDOW - prefix_tokens)
