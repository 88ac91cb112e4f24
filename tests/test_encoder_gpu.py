"""GPU tests of the HIP MiniLM encoder through the C ABI: against the committed golden vectors
(transformers.BertModel with the seed-1234 synthetic weights) and against the plain-torch fp32
restatement on fresh random batches. Tolerances: fp32 path 1e-4 absolute on unit vectors; bf16 path
cosine >= 0.999. Parity with the real all-MiniLM-L6-v2 checkpoint is UNPINNED: no
weights are available offline."""
import os

import numpy as np
import pytest
import torch

from tests import bert_ref

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
FP32_TOL = 1e-4
BF16_COS = 0.999
# harsh weight set (tests/synth_weights.py), gates just below what MI355X measured (round 3, gpurun_out/r3_harsh.txt):
#   fp32 max |diff| 2.9e-5, cosine 1.0000000 vs transformers.BertModel        -> FP32_TOL holds unchanged
#   bf16 per-text cosines 0.9839 .. 0.99997 (min 0.98390)                       -> the operand rounding of the x25 outlier channels; the round-1
#        path with f32 pre-norm sums (SHODH_ENC_UNFUSED=1) measures 0.98161: keeping the residual stream in f32 does NOT buy it back
#   INT8 vs the fp32 fixture 0.9238 .. 0.9936; the numpy restatement of the same graph scores 0.9235 .. 0.9939 against that fixture:
#        the loss is the per-tensor 8-bit quantisation of outlier-laden tensors, not this implementation
#   INT8 vs its restatement min 0.98692: two faithful evaluations of the quantised graph differ by that much here, because f32
#        reassociation flips bytes at rounding boundaries and the outliers make one byte step large (the reference's own figure for
#        pad-128 vs pad-256 of the real export is 0.9859, minilm.rs:591)
HARSH_BF16_COS = 0.975
HARSH_INT8_VS_RESTATEMENT_COS = 0.98
HARSH_INT8_VS_FP32_COS = 0.90


@pytest.fixture(scope="module")
def S():
    from shodh_memory_amd import build
    build.build()
    import shodh_memory_amd as s
    return s


@pytest.fixture(scope="module")
def ref_sd():
    from shodh_memory_amd import embedder as E
    b = E.synthetic_weights(1234)
    return {k: torch.from_numpy(v.copy()).cuda() for k, v in E.blob_to_state_dict(b).items()}


def cos(a, b):
    return (a * b).sum(1) / np.maximum(np.linalg.norm(a, axis=1) * np.linalg.norm(b, axis=1), 1e-30)


@pytest.mark.parametrize("dtype,ffn", [(0, None), (1, None), (1, "0"), (1, "1000000")])
def test_golden_vectors(S, dtype, ffn, monkeypatch):
    # ffn: bf16 forwards below SHODH_FFN_FUSED_MIN_TOKENS (default 2048) take the three-kernel feed-forward, the others the fused kernel (round 4:
    # one text is one tile on one CU for the fused kernel): "0" = always fused, "1000000" = never -- both forms against the same fixture
    if ffn is not None:
        monkeypatch.setenv("SHODH_FFN_FUSED_MIN_TOKENS", ffn)
    g = np.load(os.path.join(ROOT, "tests", "golden", "encoder_golden.npz"))
    e = S.MiniLMEmbedder(synthetic_seed=1234, dtype=dtype)
    assert e.dimension() == 384
    for name in ("b1", "b4", "edge"):
        emb = e.encode_ids(g[name + "_ids"], g[name + "_mask"])
        exp = g[name + "_emb"]
        lens = g[name + "_mask"].sum(1)
        assert (emb[lens == 0] == 0).all()                                 # empty text -> zero vector (minilm.rs:1123-1125)
        if dtype == 0:
            assert np.abs(emb - exp).max() < FP32_TOL, (name, np.abs(emb - exp).max())
        else:
            assert cos(emb[lens > 0], exp[lens > 0]).min() > BF16_COS, (name, cos(emb[lens > 0], exp[lens > 0]))
        assert np.allclose(np.linalg.norm(emb[lens > 0], axis=1), 1, atol=1e-3)


@pytest.mark.parametrize("dtype,ffn", [(0, None), (1, None), (1, "0"), (1, "1000000")])
def test_random_batches_vs_torch_reference(S, ref_sd, dtype, ffn, monkeypatch):
    if ffn is not None:
        monkeypatch.setenv("SHODH_FFN_FUSED_MIN_TOKENS", ffn)
    e = S.MiniLMEmbedder(synthetic_seed=1234, dtype=dtype)
    for b, seed in ((1, 1), (7, 2), (64, 3), (130, 4)):
        ids, mask = bert_ref.synth_batch(b, 256, seed=seed)
        with torch.no_grad():
            exp = bert_ref.encode(ref_sd, ids.cuda(), mask.cuda()).cpu().numpy()
        emb = e.encode_ids(ids.numpy().astype(np.int32), mask.numpy().astype(np.uint8))
        if dtype == 0:
            assert np.abs(emb - exp).max() < FP32_TOL, (b, np.abs(emb - exp).max())
        else:
            assert cos(emb, exp).min() > BF16_COS, (b, cos(emb, exp).min())
    t = e.stage_timings_us()
    assert t["embedding_us"] > 0 and t["tokens"] > 0


def test_full_length_and_device_api(S, ref_sd):
    e = S.MiniLMEmbedder(synthetic_seed=1234, dtype=0)
    ids, mask = bert_ref.synth_batch(3, 256, seed=9, lengths=[256, 200, 129])    # beyond the tokenizer's 128 cut: the ABI allows max_len
    with torch.no_grad():
        exp = bert_ref.encode(ref_sd, ids.cuda(), mask.cuda()).cpu().numpy()
    d_ids = ids.to(torch.int32).cuda().contiguous(); d_mask = mask.to(torch.uint8).cuda().contiguous()
    out = e.encode_ids_device(d_ids, d_mask)
    torch.cuda.synchronize()
    assert np.abs(out.cpu().numpy() - exp).max() < FP32_TOL
    from shodh_memory_amd import _lib
    bad = mask.numpy().astype(np.uint8).copy(); bad[0, 5] = 0                     # hole in the mask: rejected
    with pytest.raises(_lib.ShodhError):
        e.encode_ids(ids.numpy().astype(np.int32), bad)
    e2 = S.MiniLMEmbedder()                                                       # no weights loaded
    with pytest.raises(_lib.ShodhError):
        e2.encode_ids(ids.numpy().astype(np.int32), mask.numpy().astype(np.uint8))


def test_embedder_trait_with_a_tokenizer(S):
    """Embedder::encode / encode_query / encode_batch semantics (mod.rs:52-70, minilm.rs:1195-1377) with a
    small in-memory WordPiece tokenizer (the real tokenizer.json is not shipped)."""
    from tokenizers import Tokenizer, models, normalizers, pre_tokenizers, processors
    vocab = {"[PAD]": 0, "[UNK]": 100, "[CLS]": 101, "[SEP]": 102}
    for i, w in enumerate("the quick brown fox jumps over lazy dog memory recall vector search rust gpu hello world query passage :".split()):
        vocab[w] = 1000 + i
    tok = Tokenizer(models.WordPiece(vocab, unk_token="[UNK]"))
    tok.normalizer = normalizers.BertNormalizer(lowercase=True)
    tok.pre_tokenizer = pre_tokenizers.BertPreTokenizer()
    tok.post_processor = processors.TemplateProcessing(single="[CLS] $A [SEP]", special_tokens=[("[CLS]", 101), ("[SEP]", 102)])
    e = S.MiniLMEmbedder(tokenizer=tok, synthetic_seed=1234, dtype=0)
    a = e.encode("The quick brown fox")
    assert a.shape == (384,) and abs(np.linalg.norm(a) - 1) < 1e-4
    assert np.array_equal(e.encode_query("The quick brown fox"), a)              # symmetric model: same path
    assert not e.encode("").any()
    batch = e.encode_batch(["hello world", "", "the lazy dog"])
    assert len(batch) == 3 and not batch[1].any()
    assert np.abs(batch[0] - e.encode("hello world")).max() < 1e-5
    assert e.count_tokens("") == 2 and e.count_tokens("hello world") == 4 and e.chunk_budget_tokens() == 128
    long_text = " ".join(["fox"] * 400)
    assert e.count_tokens(long_text) == 402                                       # counted without truncation
    v = e.encode(long_text)                                                       # truncated to 128 tokens by the tokenizer
    assert abs(np.linalg.norm(v) - 1) < 1e-4 and e.stage_timings_us()["tokens"] == 128
    # asymmetric prefixes (SHODH_EMBEDDER=e5: minilm.rs:279-300)
    e5 = S.MiniLMEmbedder(tokenizer=tok, synthetic_seed=1234, dtype=0, query_prefix="query: ", doc_prefix="passage: ")
    assert not np.array_equal(e5.encode("hello world"), e5.encode_query("hello world"))
    assert e5.chunk_budget_tokens() < 128


def test_retrieval_engine_with_simplified_embedder(S):
    """retrieval.rs:2446-2535 shape: simplified (hash) embedder + deterministic vectors through
    index_memory / search_ids, chunk dedup keeps the best chunk per memory."""
    import uuid
    from tests.test_oracle_kats import ref_test_vector
    eng = S.RetrievalEngine(S.MiniLMEmbedder.new_simplified(), dimension=384, scan_mode=1)
    mids = [uuid.UUID(int=1000 + i) for i in range(25)]
    for i, m in enumerate(mids):
        eng.index_memory(m, embedding=ref_test_vector(i, 384))
    for i, m in enumerate(mids):
        res = eng.search_ids(query_embedding=ref_test_vector(i, 384), limit=3)
        assert res[0][0] == m and abs(res[0][1] - 1.0) < 1e-5 and len(res) == 3
    # a chunked memory: three vectors -> one id, max similarity kept
    big = uuid.UUID(int=5)
    eng.index_memory(big, chunks=["alpha beta", "gamma delta", "epsilon"])
    assert len(eng.id_mapping.get_vector_ids(big)) == 3
    q = eng.embedder.encode("gamma delta")
    res = eng.search_ids(query_embedding=q, limit=5)
    assert res[0][0] == big and abs(res[0][1] - 1.0) < 1e-5 and [r[0] for r in res].count(big) == 1
    assert eng.search_by_embedding(q, 5, exclude_id=big)[0][0] != big
    assert eng.search_ids() == []


def test_reference_force_quality_rebuild(S):
    """retrieval.rs:2446-2535 test_force_quality_rebuild_resets_counter_and_preserves_mapping, step for step"""
    import uuid
    from tests.test_oracle_kats import ref_test_vector
    engine = S.RetrievalEngine(S.MiniLMEmbedder.new_simplified(), dimension=384)
    ids = []
    for i in range(25):
        v = ref_test_vector(i, 384)
        vid = engine.vector_index.add_vector(v)
        mid = uuid.uuid4()
        engine.id_mapping.insert(mid, vid)
        ids.append((mid, v))
    # 24, not 25: the first add_vector seeds an empty index and is not counted as an incremental insert
    assert engine.index_health()["incremental_inserts"] == 24
    for mid, v in ids[:5]:
        assert engine.search_by_embedding(v, 3)[0][0] == mid, "top hit before rebuild must be the vector's own memory"
    engine.force_quality_rebuild()
    health = engine.index_health()
    assert health["incremental_inserts"] == 0 and health["total_vectors"] == 25
    post = engine.vector_index.extract_all_vectors()
    assert len(post) == 25
    for i, (_, v) in enumerate(ids):
        assert np.abs(post[i] - v).max() < 1e-6, "vector %d content corrupted by rebuild" % i
    for mid, v in ids[:5]:
        assert engine.search_by_embedding(v, 3)[0][0] == mid, "top hit after rebuild must be the vector's own memory"
    engine.force_quality_rebuild()                      # zero incremental inserts: a no-op, not an error
    # a soft-deleted (unmapped) vector drops out of the rebuilt index
    engine.vector_index.add_vector(ref_test_vector(30, 384))
    engine.force_quality_rebuild()
    assert engine.index_health()["total_vectors"] == 25


def test_remember_and_recall_slices(S):
    """MemorySystem::remember embed+index slice and recall's query-embed + vector leg (memory/mod.rs:1025-1051, :1088-1094,
    :1252-1256, :2873-2904, :3580-3617) over the simplified embedder."""
    import uuid
    eng = S.RetrievalEngine(S.MiniLMEmbedder.new_simplified(), dimension=384, scan_mode=1)
    mp = S.MemoryPathSlice(eng)
    texts = ["the cat sat on the mat", "a dog chased the cat", "quarterly revenue grew", "revenue grew last quarter", "gpu kernels are fun"]
    ids = [uuid.UUID(int=100 + i) for i in range(len(texts))]
    for mid, t in zip(ids, texts):
        emb, indexed, similar = mp.remember(mid, t)
        assert indexed and all(m != mid for m, _ in similar) and len(similar) <= 5         # the new memory is excluded from its own check
    assert mp.content_cache.misses == 5 and mp.content_cache.hits == 0
    emb2, _, similar = mp.remember(uuid.UUID(int=999), texts[0])                            # same content: cache hit, finds the original
    assert mp.content_cache.hits == 1 and similar[0][0] == ids[0] and abs(similar[0][1] - 1.0) < 1e-5
    res = mp.recall_vector_leg("revenue grew last quarter", max_results=2)
    assert res[0][0] == ids[3] and len(res) <= 6 and mp.query_cache.misses == 1           # vector_top_k = max_results * 3
    assert mp.recall_vector_leg("revenue grew last quarter", max_results=2) == res and mp.query_cache.hits == 1
    pre = eng.embedder.encode("gpu kernels are fun")
    assert mp.recall_vector_leg("ignored", 1, query_embedding=pre)[0][0] == ids[4] and mp.query_cache.misses == 1    # pre-computed embedding: no encode
    # polarity-sensitive + negated form: union per memory, best score kept, (score desc, id asc)
    neg = eng.embedder.encode("a dog chased the cat")
    both = mp.recall_vector_leg("the cat sat on the mat", 1, polarity_sensitive=True, negated_embedding=neg)
    top = {m for m, _ in both[:3]}
    assert len(both) <= 6 * 2 and ids[1] in top and (ids[0] in top or uuid.UUID(int=999) in top)
    assert [s for _, s in both] == sorted([s for _, s in both], reverse=True)
    assert len({m for m, _ in both}) == len(both)
    only = mp.recall_vector_leg("the cat sat on the mat", 3, episode_candidates={ids[1], ids[2]})
    assert {m for m, _ in only} <= {ids[1], ids[2]}


def test_index_memory_chunks_long_content(S):
    """retrieval.rs:646-700: content beyond the 128-token window is split by the structural chunker, one vector per chunk, all
    mapped to the memory; a query for text from the END of the content still finds it (the reason the chunker exists)."""
    import uuid
    from shodh_memory_amd.chunking import ChunkConfig, chunk_text
    eng = S.RetrievalEngine(S.MiniLMEmbedder.new_simplified(), dimension=384, scan_mode=1)
    long_text = " ".join("Paragraph %d talks about topic number %d in some detail." % (i, i) for i in range(120)) + " The launch code is heliotrope-seven."
    short = uuid.UUID(int=1)
    eng.index_memory(short, content="a short note about cats")
    big = uuid.UUID(int=2)
    ids = eng.index_memory(big, content=long_text)
    expect = chunk_text(long_text, ChunkConfig.for_budget(eng.embedder.chunk_budget_tokens()), eng.embedder.count_tokens)
    assert expect.was_chunked and len(ids) == len(expect.chunks) > 3 and eng.id_mapping.get_vector_ids(big) == ids
    assert all(eng.embedder.count_tokens(c) <= 128 for c in expect.chunks)
    assert len(eng.id_mapping.get_vector_ids(short)) == 1
    res = eng.search_ids(query_embedding=eng.embedder.encode(expect.chunks[-1]), limit=2)
    assert res[0][0] == big and abs(res[0][1] - 1.0) < 1e-5 and [r[0] for r in res].count(big) == 1
    eng.index_memory(big, content="now it is short")                         # re-index: the old chunk vectors lose their mapping
    assert len(eng.id_mapping.get_vector_ids(big)) == 1 and all(eng.id_mapping.get_memory_id(v) is None for v in ids)


def test_big_batches_run_as_sub_batches_with_identical_results(S):
    """more than 8192 texts in one call are encoded as sub-batches inside the library (texts are independent in bf16 / fp32):
    same bits as encoding the pieces one by one"""
    from shodh_memory_amd import _lib as L
    rng = np.random.default_rng(3)
    b, ML = 8192 + 700, 256
    lens = rng.integers(1, 40, b)
    ids = np.zeros((b, ML), np.int32); mask = np.zeros((b, ML), np.uint8)
    for i, n in enumerate(lens):
        ids[i, :n] = rng.integers(1000, 30000, n); mask[i, :n] = 1
    enc = S.MiniLMEmbedder(synthetic_seed=77, dtype=L.DTYPE_BF16)
    whole = enc.encode_ids(ids, mask)
    parts = np.concatenate([enc.encode_ids(ids[:8192], mask[:8192]), enc.encode_ids(ids[8192:], mask[8192:])])
    assert whole.shape == (b, 384) and whole.tobytes() == parts.tobytes()
    assert np.allclose(np.linalg.norm(whole, axis=1), 1.0, atol=1e-3)


def test_harsh_weight_set_fp32_bf16_int8(S):
    """a second weight set with outlier channels (x25), LayerNorm gains over [0.1, 10] and a heavy-tailed word table
    (tests/synth_weights.py), golden vectors from transformers.BertModel in fp32 (tests/golden/encoder_harsh_golden.npz). The cosines are
    MEASURED and printed; the gates sit just below what was measured on MI355X (stated in DESIGN.md), not at an assumed 0.999."""
    from oracle import int8_ref as R
    from shodh_memory_amd import _lib as L
    from shodh_memory_amd import embedder as E
    from tests import synth_weights as SW
    g = np.load(os.path.join(ROOT, "tests", "golden", "encoder_harsh_golden.npz"))
    blob = SW.harsh_blob(E, int(g["seed"]))
    ids, mask, exp = g["h8_ids"], g["h8_mask"], g["h8_emb"]
    e32 = S.MiniLMEmbedder(weights=blob, dtype=L.DTYPE_FP32)
    a32 = e32.encode_ids(ids, mask)
    d32 = float(np.abs(a32 - exp).max())
    c32 = cos(a32, exp)
    e16 = S.MiniLMEmbedder(weights=blob, dtype=L.DTYPE_BF16)
    c16 = cos(e16.encode_ids(ids, mask), exp)
    e8 = S.MiniLMEmbedder(weights=blob, dtype=L.DTYPE_INT8)
    a8 = e8.encode_ids(ids, mask)
    c8 = cos(a8, exp)
    r8 = R.encode(E.blob_to_state_dict(blob), ids, mask)               # the restated INT8 graph on the same weights (self-quantised: per tensor, symmetric)
    c8r = cos(a8, r8)
    print("harsh weights: fp32 max|diff| %.3g min cos %.7f | bf16 min cos %.5f | int8 vs fp32 fixture min cos %.5f | int8 vs its restatement min cos %.6f"
          % (d32, c32.min(), c16.min(), c8.min(), c8r.min()))
    print("harsh per-text cosines bf16", np.round(c16, 5), "int8", np.round(c8, 5), "restated int8 vs fp32 fixture", np.round(cos(r8, exp), 5))
    assert d32 < 2e-4 and c32.min() > 0.999999, (d32, c32)
    assert c16.min() > HARSH_BF16_COS, c16
    assert c8r.min() > HARSH_INT8_VS_RESTATEMENT_COS, c8r
    assert c8.min() > HARSH_INT8_VS_FP32_COS, c8
