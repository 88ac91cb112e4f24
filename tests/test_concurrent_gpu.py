"""GPU tests of the reference's real call pattern (round 5): MANY callers, ONE item each, on one handle.

`recall` is one query per call under a read lock (src/handlers/recall.rs:512-513 spawn_blocking + memory.read(); src/memory/retrieval.rs:912-918
vector_index.read()), `remember` one encode() behind Mutex<Session> (src/embeddings/minilm.rs:889-897). The library coalesces such calls
(csrc/combiner.h): calls that arrive while a device pass is in flight share the next pass. The front may change WHEN work runs, never WHAT comes
back: every caller must get the bytes its own solo call produces -- for searches (flat MFMA / exact / IVF-PQ / sharded, mixed k), for encodes
(INT8 per text, bf16, fp32) and for the chunked `index_memory` path that depends on it."""
import threading
import uuid

import numpy as np
import pytest

from tests import synth

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def S():
    from shodh_memory_amd import build
    build.build()
    import shodh_memory_amd as s
    return s


def _solo(idx, q, k):
    """every query alone, coalescing off: the reference answers the concurrent runs are compared with"""
    idx.set_coalesce(False)
    ids = np.empty((q.shape[0], k), np.uint32); dist = np.empty((q.shape[0], k), np.float32)
    for i in range(q.shape[0]):
        a, b, _ = idx.search_batch(q[i:i + 1], k)
        ids[i], dist[i] = a[0], b[0]
    idx.set_coalesce(True)
    return ids, dist


@pytest.mark.parametrize("scan_mode,n_rows", [(2, 200_000), (1, 60_000)])
@pytest.mark.parametrize("k", [10, 120])
def test_concurrent_single_query_callers_get_their_solo_bytes(S, oracle, scan_mode, n_rows, k):
    from shodh_memory_amd import _lib as L
    from tools import callers
    q = synth.queries(96)
    rows = synth.corpus(n_rows, queries=q)
    idx = S.VamanaIndex(S.VamanaConfig(dimension=384, scan_mode=scan_mode))
    idx.build(rows)
    idx.mark_deleted_many(np.arange(0, n_rows, 17, dtype=np.uint32))          # tombstones travel through the shared pass too
    e_ids, e_dist = _solo(idx, q, k)
    dmask = np.zeros(n_rows, np.uint8); dmask[::17] = 1
    o_ids, o_dist = oracle.brute_force_batch(rows, q[:8], k, deleted=dmask)
    assert np.array_equal(e_ids[:8], o_ids) and e_dist[:8].tobytes() == o_dist.tobytes()          # the solo answers are the oracle's
    idx.coalesce_stats(reset=True)
    for threads in (2, 16, 48):
        r = callers.search(L.lib(), idx.handle, q, k, threads=threads, calls_per_thread=24, expect=(e_ids, e_dist))
        assert r.errors == 0 and r.mismatches == 0, "threads %d: %d of %d concurrent results differ from the solo call" % (threads, r.mismatches, r.calls)
    st = idx.coalesce_stats()
    print("scan_mode %d k %d: %d calls in %d passes, largest pass %d" % (scan_mode, k, st["calls"], st["passes"], st["largest"]))
    assert st["largest"] > 1, "no two calls ever shared a pass"
    idx.close()


def test_first_search_on_a_fresh_workspace_under_load(S):
    """64 threads make their FIRST call at the same moment (coalescing off: every caller creates its own workspace while the device is busy with the
    others). The workspace's survivor counter is cleared with a memset that has to be complete before the first kernel reads it -- round 5 found one
    wrong list in 6400 such calls (bench.py concurrent_callers, coalesce off), a race that had been there since the single-query scan was written."""
    from shodh_memory_amd import _lib as L
    from tools import callers
    q = synth.queries(64)
    rows = synth.corpus(150_000, queries=q)
    ref = S.VamanaIndex(S.VamanaConfig(dimension=384)); ref.build(rows)
    e120 = _solo(ref, q, 120)
    ref.close()
    for rep in range(5):
        idx = S.VamanaIndex(S.VamanaConfig(dimension=384)); idx.build(rows)
        idx.set_coalesce(False)
        r = callers.search(L.lib(), idx.handle, q, 120, threads=64, calls_per_thread=6, warmup=0, expect=e120)
        assert r.errors == 0 and r.mismatches == 0, "fresh index %d: %d of %d first calls wrong" % (rep, r.mismatches, r.calls)
        idx.close()


def test_searches_coalesce_while_the_index_grows(S):
    """readers and a writer on one handle (the reference: recall under vector_index.read(), remember under .write(), retrieval.rs:680-712, :912): 12 threads
    search one query per call while another thread appends rows batch by batch. A pass runs under the index' shared lock, an append under the exclusive
    one, so every answer must be the exact top-k of SOME prefix the index went through -- never a mixture, never a torn list."""
    n0, step, n_steps, k = 60_000, 5_000, 8, 10
    q = synth.queries(32)
    rows = synth.corpus(n0 + step * n_steps, queries=q)
    states = []
    ref = S.VamanaIndex(S.VamanaConfig(dimension=384))
    ref.build(rows[:n0])
    for i in range(n_steps + 1):
        if i:
            ref.add_vectors(rows[n0 + (i - 1) * step:n0 + i * step])
        ids, dist, _ = ref.search_batch(q, k)
        states.append((ids.copy(), dist.copy()))
    ref.close()
    idx = S.VamanaIndex(S.VamanaConfig(dimension=384))
    idx.build(rows[:n0])
    bad, seen_states = [], set()
    stop = threading.Event()

    def reader(t):
        i = t
        while not stop.is_set():
            j = i % 32
            ids, dist, _ = idx.search_batch(q[j:j + 1], k)
            hit = [s for s in range(n_steps + 1) if np.array_equal(ids[0], states[s][0][j]) and dist[0].tobytes() == states[s][1][j].tobytes()]
            if not hit:
                bad.append((t, j))
            else:
                seen_states.add(hit[-1])
            i += 12
    th = [threading.Thread(target=reader, args=(t,)) for t in range(12)]
    for x in th: x.start()
    for i in range(1, n_steps + 1):
        idx.add_vectors(rows[n0 + (i - 1) * step:n0 + i * step])
    stop.set()
    for x in th: x.join()
    assert not bad, bad[:5]
    ids, dist, _ = idx.search_batch(q, k)
    assert np.array_equal(ids, states[-1][0]) and dist.tobytes() == states[-1][1].tobytes()
    print("index states seen by the readers while it grew:", sorted(seen_states), idx.coalesce_stats())
    idx.close()


def test_mixed_k_and_small_batches_share_a_pass(S):
    """members with different k (a recall at limit 10 asks k = 120, a dedup lookup k = 5) and calls with a few queries: each gets exactly its own answer"""
    n = 120_000
    q = synth.queries(64)
    rows = synth.corpus(n, queries=q)
    idx = S.VamanaIndex(S.VamanaConfig(dimension=384))
    idx.build(rows)
    shapes = [(1, 5), (1, 120), (3, 10), (1, 1), (4, 300), (2, 40)]          # (queries per call, k)
    idx.set_coalesce(False)
    expect = {}
    for t, (nq, k) in enumerate(shapes):
        for rep in range(6):
            s0 = (t * 11 + rep * 5) % (64 - nq)
            expect[(t, rep)] = idx.search_batch(q[s0:s0 + nq], k)
    idx.set_coalesce(True)
    idx.coalesce_stats(reset=True)
    bad = []
    start = threading.Barrier(len(shapes) * 2)

    def work(t, nq, k):
        start.wait()
        for rep in range(6):
            s0 = (t * 11 + rep * 5) % (64 - nq)
            ids, dist, cnt = idx.search_batch(q[s0:s0 + nq], k)
            e = expect[(t, rep)]
            if not (np.array_equal(ids, e[0]) and dist.tobytes() == e[1].tobytes() and np.array_equal(cnt, e[2])):
                bad.append((t, rep))
    th = [threading.Thread(target=work, args=(t, nq, k)) for t, (nq, k) in enumerate(shapes)] * 1
    th += [threading.Thread(target=work, args=(t, nq, k)) for t, (nq, k) in enumerate(shapes)]
    for x in th: x.start()
    for x in th: x.join()
    assert not bad, bad
    print("mixed k:", idx.coalesce_stats())
    idx.close()


def test_concurrent_callers_on_ivfpq_and_sharded_indexes(S):
    from shodh_memory_amd import _lib as L
    from shodh_memory_amd.distributed import MultiGpuIndex
    from tools import callers
    q = synth.queries(48)
    rows = synth.corpus(40_000, queries=q)
    # IVF-PQ (SpannIndex::search is one query per call as well)
    sp = S.SpannIndex(dimension=384, num_probes=8)
    sp.build(rows[:20_000], num_partitions=64, kmeans_iterations=4, pq_iterations=3, seed=5)
    e_ids, e_dist = _solo(sp, q, 10)
    sp.coalesce_stats(reset=True)
    r = callers.search(L.lib(), sp.handle, q, 10, threads=12, calls_per_thread=16, expect=(e_ids, e_dist))
    assert r.errors == 0 and r.mismatches == 0
    print("ivfpq:", sp.coalesce_stats())
    assert sp.coalesce_stats()["largest"] > 1
    sp.close()
    # sharded index, two shards on this GPU: concurrent host searches share one pass over the shards and one exchange; several calls in flight
    mg = MultiGpuIndex([0, 0], dim=384, block_log2=10)
    mg.build(rows)
    one = S.VamanaIndex(S.VamanaConfig(dimension=384)); one.build(rows)
    for k in (10, 120):
        o_ids, o_dist = _solo(one, q, k)
        mg.set_coalesce(False)
        s_ids = np.empty_like(o_ids); s_dist = np.empty_like(o_dist)
        for i in range(q.shape[0]):
            a, b, _ = mg.search_batch(q[i:i + 1], k); s_ids[i], s_dist[i] = a[0], b[0]
        assert np.array_equal(s_ids, o_ids) and s_dist.tobytes() == o_dist.tobytes()          # sharded == one index (bit-identical merge)
        # coalescing off: the calls run CONCURRENTLY on their own slots (no whole-call mutex any more) and still get their own answers
        r = callers.search(L.lib(), mg._h, q, k, threads=8, calls_per_thread=12, expect=(o_ids, o_dist), sharded=True)
        assert r.errors == 0 and r.mismatches == 0
        mg.set_coalesce(True)
        mg.coalesce_stats(reset=True)
        r = callers.search(L.lib(), mg._h, q, k, threads=16, calls_per_thread=12, expect=(o_ids, o_dist), sharded=True)
        assert r.errors == 0 and r.mismatches == 0
        print("sharded k %d:" % k, mg.coalesce_stats())
        assert mg.coalesce_stats()["largest"] > 1
    one.close(); mg.close()


def _tokens(n, seed, vocab=30522):
    rng = np.random.default_rng(seed)
    ids = np.zeros((n, 256), np.int32); mask = np.zeros((n, 256), np.uint8)
    lens = rng.integers(3, 129, n)
    lens[0] = 128; lens[1] = 3
    for i, ln in enumerate(lens):
        ids[i, :ln] = rng.integers(1000, vocab, ln); ids[i, 0] = 101; ids[i, ln - 1] = 102; mask[i, :ln] = 1
    return ids, mask


@pytest.mark.parametrize("dtype", [2, 1, 0])          # INT8 (the reference's default model), bf16, fp32
def test_a_text_has_one_embedding_whoever_shares_its_forward(S, dtype):
    """encode(t) == encode_each([t] + mates)[i] == the vector a concurrent caller gets when its call is coalesced with others -- BYTES, for all three
    dtypes (fp32 / bf16: the kernel forms of a one-text call are chosen by a batch-independent rule; INT8: quant_scope PER_TEXT)."""
    from shodh_memory_amd import _lib as L
    from tools import callers
    e = S.MiniLMEmbedder(synthetic_seed=1234, dtype=dtype)
    n = 40
    ids, mask = _tokens(n, seed=3)
    e.set_coalesce(False)
    solo = np.concatenate([e.encode_ids(ids[i:i + 1], mask[i:i + 1]) for i in range(n)], 0)
    e.set_coalesce(True)
    assert np.abs(np.linalg.norm(solo, axis=1) - 1).max() < 1e-3
    each = e.encode_ids(ids, mask, scope=L.QUANT_SCOPE_PER_TEXT)
    assert each.tobytes() == solo.tobytes(), "encode_each differs from N x encode (dtype %d): max |diff| %g" % (dtype, np.abs(each - solo).max())
    perm = np.array([7, 0, 39, 12, 1])
    assert e.encode_ids(ids[perm], mask[perm], scope=L.QUANT_SCOPE_PER_TEXT).tobytes() == solo[perm].tobytes()       # any mates, any position
    # a big batch: 2600 texts (~170 k tokens, beyond the size where the batch call switches kernel forms) still gives every text its solo bytes
    big = np.tile(np.arange(n), 65)
    out = e.encode_ids(ids[big], mask[big], scope=L.QUANT_SCOPE_PER_TEXT)
    assert out.tobytes() == solo[big].tobytes()
    # concurrent one-text callers, coalesced
    e.coalesce_stats(reset=True)
    for threads in (2, 12, 32):
        r = callers.encode(L.lib(), e._h, ids, mask, 384, threads=threads, calls_per_thread=10, expect=solo)
        assert r.errors == 0 and r.mismatches == 0, "dtype %d threads %d: %d of %d coalesced embeddings differ from the solo call" % (dtype, threads, r.mismatches, r.calls)
    st = e.coalesce_stats()
    print("dtype %d: %d encode calls in %d forwards, largest %d" % (dtype, st["calls"], st["passes"], st["largest"]))
    assert st["largest"] > 1
    # calls with several texts run next to the coalesced ones on their own scratch sets (no forward mutex): both kinds keep their answers
    batch_ref = e.encode_ids(ids[:16], mask[:16])
    bad = []

    def batches():
        for _ in range(6):
            if e.encode_ids(ids[:16], mask[:16]).tobytes() != batch_ref.tobytes():
                bad.append("batch")

    def singles(t):
        for j in range(12):
            i = (t * 5 + j) % n
            if e.encode_ids(ids[i:i + 1], mask[i:i + 1]).tobytes() != solo[i:i + 1].tobytes():
                bad.append(("single", i))
    th = [threading.Thread(target=batches) for _ in range(2)] + [threading.Thread(target=singles, args=(t,)) for t in range(6)]
    for x in th: x.start()
    for x in th: x.join()
    assert not bad, bad[:5]
    e.close()


def test_one_text_int8_graph_replay_equals_plain_launches(S, monkeypatch):
    """encode() of ONE text on the INT8 model replays a captured hipGraph (encoder.hip: encode_one_graph). Same kernels, same arguments: the vectors are
    byte-equal to plain launches (SHODH_ENC_GRAPH=0), also when other forwards use the same scratch set in between (they overwrite the token maps the
    graph reads and may reallocate what it points at) and for texts the graph does not take (more than 128 tokens)."""
    from shodh_memory_amd import _lib as L
    n = 24
    ids, mask = _tokens(n, seed=8)
    ids[5, :200] = np.arange(1000, 1200); ids[5, 0] = 101; ids[5, 199] = 102; mask[5, :200] = 1          # beyond the tokenizer's window: plain path
    monkeypatch.setenv("SHODH_ENC_GRAPH", "0")
    plain = S.MiniLMEmbedder(synthetic_seed=1234, dtype=L.DTYPE_INT8)
    monkeypatch.setenv("SHODH_ENC_GRAPH", "1")
    monkeypatch.setenv("SHODH_ENC_SLOTS", "1")          # one scratch set: every call below lands on the one that holds the graph
    graph = S.MiniLMEmbedder(synthetic_seed=1234, dtype=L.DTYPE_INT8)
    ref = np.concatenate([plain.encode_ids(ids[i:i + 1], mask[i:i + 1]) for i in range(n)], 0)
    got = np.concatenate([graph.encode_ids(ids[i:i + 1], mask[i:i + 1]) for i in range(n)], 0)
    assert got.tobytes() == ref.tobytes()
    b_ref = plain.encode_ids(ids[:7], mask[:7])
    for rep in range(3):
        assert graph.encode_ids(ids[:7], mask[:7]).tobytes() == b_ref.tobytes()                         # a batch call in between (same scratch set)
        big = np.tile(np.arange(n), 40 * (rep + 1))
        graph.encode_ids(ids[big], mask[big], scope=L.QUANT_SCOPE_PER_TEXT)                             # ... and one that makes the scratch set grow
        again = np.concatenate([graph.encode_ids(ids[i:i + 1], mask[i:i + 1]) for i in range(n)], 0)
        assert again.tobytes() == ref.tobytes()
    plain.close(); graph.close()


def _word_tokenizer():
    from tokenizers import Tokenizer, models, pre_tokenizers, processors
    words = ["[PAD]", "[UNK]", "[CLS]", "[SEP]"] + ["w%d" % i for i in range(400)]
    tok = Tokenizer(models.WordLevel({w: i for i, w in enumerate(words)}, unk_token="[UNK]"))
    tok.pre_tokenizer = pre_tokenizers.Whitespace()
    tok.post_processor = processors.TemplateProcessing(single="[CLS] $A [SEP]", special_tokens=[("[CLS]", 2), ("[SEP]", 3)])
    return tok


@pytest.mark.parametrize("dtype", [2, 1])
def test_index_memory_embeds_chunks_one_by_one(S, dtype):
    """RetrievalEngine.index_memory of a chunked text (src/memory/retrieval.rs:668-676: one encode() per chunk): the stored vectors are byte-equal to
    [encode(c) for c in chunks]. With the INT8 model (the reference's default) that is NOT what encode_batch(chunks) gives -- its ranges span all
    chunks (retrieval.py used encode_batch here)."""
    pytest.importorskip("tokenizers")
    from shodh_memory_amd.chunking import ChunkConfig, chunk_text
    from shodh_memory_amd.retrieval import RetrievalEngine
    e = S.MiniLMEmbedder(tokenizer=_word_tokenizer(), synthetic_seed=1234, dtype=dtype)
    eng = RetrievalEngine(e, dimension=384)
    rng = np.random.default_rng(1)
    sentences = [" ".join("w%d" % w for w in rng.integers(0, 400, rng.integers(8, 30))) + "." for _ in range(60)]
    content = " ".join(sentences)
    r = chunk_text(content, ChunkConfig.for_budget(e.chunk_budget_tokens()), e.count_tokens)
    assert r.was_chunked and len(r.chunks) >= 3
    mid = uuid.UUID(int=7)
    vids = eng.index_memory(mid, content=content)
    assert len(vids) == len(r.chunks)
    stored = eng.vector_index.extract_all_vectors()
    one_by_one = np.stack([e.encode(c) for c in r.chunks])
    assert stored.tobytes() == one_by_one.tobytes()
    if dtype == 2:
        batch = np.stack(e.encode_batch(r.chunks))
        assert batch.tobytes() != one_by_one.tobytes()                # (the function the round-4 code computed here)
    # the memory is found through any of its chunks, scored by its best chunk (retrieval.rs:927-961)
    hits = eng.search_ids(query_text=r.chunks[1], limit=3)
    assert hits and hits[0][0] == mid and hits[0][1] > 0.999
    e.close()


def test_configs2_chained_int8_per_text_encode_add_recall(S, oracle):
    """configs[2] in the reference's DEFAULT dtype: texts -> INT8 encode, per text (N x `remember`, memory/mod.rs:1037) -> insert -> recall, on the
    device. The stored vectors are what N single encode() calls give (spot-checked byte for byte), recall over them is bit-equal to the oracle."""
    import torch
    from shodh_memory_amd import _lib as L
    n_texts, bsz, nq, k = 12_000, 2048, 32, 10
    e = S.MiniLMEmbedder(synthetic_seed=1234, dtype=L.DTYPE_INT8, quant_scope=L.QUANT_SCOPE_PER_TEXT)
    idx = S.VamanaIndex(S.VamanaConfig(dimension=384, reserve_rows=n_texts))
    ids, mask = _tokens(n_texts, seed=21)
    d_ids = torch.from_numpy(ids).cuda(); d_mask = torch.from_numpy(mask).cuda()
    emb = torch.empty((bsz, 384), dtype=torch.float32, device="cuda")
    for b0 in range(0, n_texts, bsz):
        b = min(bsz, n_texts - b0)
        e.encode_ids_device(d_ids[b0:b0 + b].contiguous(), d_mask[b0:b0 + b].contiguous(), out=emb[:b])
        torch.cuda.synchronize()
        idx.add_vectors(emb[:b])
    assert idx.len() == n_texts
    rows = idx.extract_all_vectors()
    pick = np.linspace(0, n_texts - 1, nq).astype(np.int64)
    singles = np.concatenate([e.encode_ids(ids[i:i + 1], mask[i:i + 1]) for i in pick], 0)        # `recall`'s own encode() of the same texts
    assert singles.tobytes() == rows[pick].tobytes(), "a text's INT8 embedding depends on its ingest batch"
    got_ids, got_dist, counts = idx.search_batch(singles, k)
    exp_ids, exp_dist = oracle.brute_force_batch(rows, singles, k)
    assert (counts == k).all() and np.array_equal(got_ids, exp_ids) and got_dist.tobytes() == exp_dist.tobytes()
    assert (got_dist[:, 0] < -0.999).all()                           # self (or a duplicate text) is the top hit
    # the recall side one query per call, k = 120 (retrieval.rs:913-918), from 8 threads: the same lists
    from tools import callers
    e120 = _solo(idx, singles, 120)
    r = callers.search(L.lib(), idx.handle, singles, 120, threads=8, calls_per_thread=8, expect=e120)
    assert r.errors == 0 and r.mismatches == 0
    e.close(); idx.close()
