"""Host-side glue exported by the C ABI (no device): hash embedder, search_ids post-processing and
RRF must equal the oracle's restatement bit-for-bit."""
import ctypes as C
import os
import uuid

import numpy as np
import pytest


@pytest.fixture(scope="module")
def L():
    from shodh_memory_amd import build
    build.build()
    from shodh_memory_amd import _lib
    _lib.lib()
    return _lib


def test_hash_embed_matches_oracle(L, oracle):
    texts = ["Hello world", "", "a", "ab", "  leading and   trailing  ", "café olé 😀 naïve", "x" * 500,
             "The quick brown fox jumps over the lazy dog " * 5, "tab\tnew\nline　ideographic space"]
    for t in texts:
        b = t.encode("utf-8")
        for dim in (384, 128, 1024):
            out = np.zeros(dim, np.float32)
            assert L.lib().shodh_hash_embed(b, len(b), dim, out.ctypes.data) == 0
            assert out.tobytes() == oracle.hash_embed(t, dim).tobytes(), (t[:20], dim)
    v = np.zeros(384, np.float32)
    L.lib().shodh_hash_embed(b"Hello world", 11, 384, v.ctypes.data)          # minilm.rs:1399-1417
    assert abs(float(np.linalg.norm(v)) - 1.0) < 1e-5


def test_search_ids_postprocess_matches_oracle(L, oracle):
    rng = np.random.default_rng(4)
    n_vec = 500
    mem = [uuid.UUID(int=int(x)).bytes for x in rng.integers(1, 2 ** 62, 120)]
    v2m = np.frombuffer(b"".join(mem[int(i)] if rng.random() > 0.1 else b"\xff" * 16 for i in rng.integers(0, 120, n_vec)), np.uint8).reshape(-1, 16).copy()
    for trial in range(20):
        n = int(rng.integers(0, 200))
        vec_ids = rng.integers(0, n_vec + 20, n).astype(np.uint32)
        dists = (-rng.random(n)).astype(np.float32)
        dists[rng.random(n) < 0.2] = np.float32(-0.5)                       # ties -> uuid order decides
        limit = int(rng.integers(1, 40))
        ou = np.zeros((limit, 16), np.uint8); os_ = np.zeros(limit, np.float32)
        m = L.lib().shodh_search_ids_postprocess(vec_ids.ctypes.data, dists.ctypes.data, n, v2m.ctypes.data, n_vec, limit, ou.ctypes.data, os_.ctypes.data)
        eu, es = oracle.search_ids_postprocess(vec_ids, dists, v2m, limit)
        assert m == len(eu) and ou[:m].tobytes() == eu.tobytes() and os_[:m].tobytes() == es.tobytes()


def test_rrf_matches_oracle_and_reference_kats(L, oracle):
    id1, id2, id3 = (uuid.UUID(int=i).bytes for i in (1, 2, 3))

    def fuse(k, w, lists):
        w = np.asarray(w, np.float32)
        flat = np.frombuffer(b"".join(b"".join(l) for l in lists) or b"\0" * 16, np.uint8).copy()
        lens = (C.c_size_t * len(lists))(*[len(l) for l in lists])
        cap = max(sum(len(l) for l in lists), 1)
        ou = np.zeros((cap, 16), np.uint8); os_ = np.zeros(cap, np.float32)
        m = L.lib().shodh_rrf_fuse(k, w.ctypes.data, len(lists), flat.ctypes.data, lens, ou.ctypes.data, os_.ctypes.data, cap)
        return [bytes(ou[i]) for i in range(m)], os_[:m].copy()
    ids, sc = fuse(60.0, [0.5, 0.5], [[id1, id2, id3], [id2, id1, id3]])     # hybrid_search.rs:970-1029
    assert abs(dict(zip(ids, sc))[id1] - dict(zip(ids, sc))[id2]) < 1e-4 and ids[2] == id3
    lo, hi = uuid.UUID(int=5).bytes, uuid.UUID(int=9).bytes
    ids, sc = fuse(60.0, [0.5, 0.5], [[hi], [lo]])                           # :1031-1056 ties -> lower MemoryId first
    assert ids == [lo, hi]
    rng = np.random.default_rng(0)
    pool = [uuid.UUID(int=int(x)).bytes for x in rng.integers(1, 2 ** 60, 50)]
    for _ in range(10):
        lists = [[pool[int(i)] for i in rng.permutation(50)[:int(rng.integers(0, 30))]] for _ in range(3)]
        w = rng.random(3).astype(np.float32)
        a_ids, a_sc = fuse(60.0, w, lists)
        e_ids, e_sc = oracle.rrf_fuse(60.0, w, lists)
        assert a_ids == e_ids and a_sc.tobytes() == e_sc.tobytes()


def test_reference_partition_count(oracle):
    """spann.rs:1188-1195 test_partition_count: ceil(sqrt(n)), through the Python mirror (no device needed) and the oracle"""
    from shodh_memory_amd.index import SpannIndex
    for n, p in ((100, 10), (10000, 100), (1000000, 1000), (1, 1), (0, 1), (2000, 45), (101, 11)):
        assert SpannIndex.compute_partitions(n) == p
        assert oracle.spann_compute_partitions(n) == p


def test_reference_id_mapping_tests():
    """retrieval.rs:2152-2197, :2361-2428: the reference's IdMapping unit tests against the Python mirror"""
    from shodh_memory_amd.retrieval import IdMapping
    new_id = uuid.uuid4
    m, mem = IdMapping(), new_id()                               # test_id_mapping_basic
    m.insert(mem, 42)
    assert m.len() == 1 and m.get_memory_id(42) == mem
    m, mem = IdMapping(), new_id()                               # test_id_mapping_chunks
    m.insert_chunks(mem, [1, 2, 3])
    assert m.len() == 1 and [m.get_memory_id(i) for i in (1, 2, 3)] == [mem] * 3
    removed = m.remove_all(mem)                                  # test_id_mapping_remove_all
    assert len(removed) == 3 and m.len() == 0 and m.get_memory_id(1) is None and m.remove_all(mem) == []
    m = IdMapping()                                              # test_id_mapping_clear
    m.insert(new_id(), 1); m.insert(new_id(), 2)
    m.clear()
    assert m.len() == 0
    m, mem = IdMapping(), new_id()                               # test_id_mapping_insert_is_idempotent
    m.insert(mem, 10)
    assert m.len() == 1 and m.get_memory_id(10) == mem
    m.insert(mem, 20)
    assert m.len() == 1 and m.get_memory_id(20) == mem and m.get_memory_id(10) is None, "old vector_id should be removed to prevent orphan"
    assert m.memory_to_vectors[mem] == [20]
    m, mem = IdMapping(), new_id()                               # test_id_mapping_insert_chunks_is_idempotent
    m.insert_chunks(mem, [1, 2, 3])
    assert m.len() == 1 and m.memory_to_vectors[mem] == [1, 2, 3]
    m.insert_chunks(mem, [10, 11])
    assert m.len() == 1 and m.memory_to_vectors[mem] == [10, 11]
    assert all(m.get_memory_id(i) is None for i in (1, 2, 3)) and m.get_memory_id(10) == mem and m.get_memory_id(11) == mem
    m, m1, m2 = IdMapping(), new_id(), new_id()                  # test_id_mapping_vector_count_stable_after_reinsert
    m.insert(m1, 1); m.insert(m2, 2)
    assert len(m.vector_to_memory) == 2
    m.insert(m1, 3)
    assert len(m.vector_to_memory) == 2, "vector_to_memory should not grow on re-insert"


def test_ranking_tail_matches_oracle_and_reference_kats(L, oracle):
    """relevance.rs:680-705, :1524-1547, :801-918 through the C ABI: the reference's own unit-test values, then random
    candidate sets against the oracle's restatement (bit-for-bit: both go through the same libm expf / log2f)."""
    import shodh_memory_amd as M
    # relevance.rs tests (test_tag_score_*, recency): same literals as tests/test_oracle_kats.py::test_tag_score_and_recency
    assert M.calculate_tag_score("I love Rust programming", ["rust"]) == 1.0
    assert M.calculate_tag_score("Learning Rust", ["rust", "python"]) == 0.5
    assert M.calculate_tag_score("Hello world", ["rust"]) == 0.0
    assert M.calculate_tag_score("Test", []) == 0.0
    assert M.calculate_tag_score("deploy kube clusters", ["kubernetes", "Deploy"]) == 1.0      # tag starts with a context word; case folded
    assert M.apply_recency_boost(0.5, 0, 24, 1.2) > 0.5
    assert abs(M.apply_recency_boost(0.5, 48, 24, 1.2) - 0.5) < 1e-3
    assert M.apply_recency_boost(0.5, -3, 24, 1.2) == 0.5 and M.apply_recency_boost(0.95, 0, 24, 1.2) == 1.0
    for ctx, tags in (("I love Rust programming", ["rust"]), ("a  b\tc", ["B", "zz", "c"]), ("naïve café", ["caf", "x"])):
        assert np.float32(M.calculate_tag_score(ctx, tags)) == np.float32(oracle.calculate_tag_score(ctx, tags))
    cfg0 = L.RelevanceCfg()
    L.lib().shodh_relevance_cfg_default(C.byref(cfg0))                    # test_relevance_config_defaults (:1795-1802) + :147-177
    assert (cfg0.max_results, cfg0.recency_boost_hours) == (5, 24)
    assert (np.float32(cfg0.min_importance), np.float32(cfg0.recency_boost_multiplier), np.float32(cfg0.graph_boost_multiplier)) == (np.float32(0.3), np.float32(1.2), np.float32(1.15))
    rng = np.random.default_rng(8)
    w = M.LearnedWeights()
    for trial in range(30):
        n = int(rng.integers(0, 40))
        cands = []
        for i in range(n):
            cands.append(dict(semantic=float(np.float32(rng.random())) if rng.random() < 0.8 else 0.0,
                              entity=float(np.float32(rng.random())) if rng.random() < 0.5 else 0.0, tag=float(np.float32(rng.integers(0, 4) / 3)),
                              importance=float(np.float32(rng.random())), momentum=float(np.float32(rng.uniform(-1, 1))), access_count=int(rng.integers(0, 40)),
                              graph_strength=float(np.float32(rng.random())), age_hours=int(rng.integers(-5, 100)),
                              created_at_ns=int(rng.integers(0, 4)) * 10 ** 9, uuid=uuid.UUID(int=int(rng.integers(1, 2 ** 62))).bytes))
        if n > 4:                                                # exact score ties: created_at desc, then id asc decide
            for j in (1, 2, 3):
                cands[j] = dict(cands[0], created_at_ns=cands[0]["created_at_ns"] + (j % 2), uuid=uuid.UUID(int=j).bytes)
        cfg = dict(max_results=int(rng.integers(1, 8)), min_importance=0.3, recency_boost_hours=int(rng.choice([0, 24, 72])))
        got = M.rank_surfaced(w, cands, **cfg)
        exp = oracle.rank_surfaced(oracle_weights(oracle), cands, **cfg)
        assert [(g[0], np.float32(g[1]).tobytes(), g[2]) for g in got] == [(e[0], np.float32(e[1]).tobytes(), M.relevance.REASONS[e[2]]) for e in exp], trial
        sc = [g[1] for g in got]
        assert sc == sorted(sc, reverse=True) and all(s >= 0.25 for s in sc) and len(got) <= cfg["max_results"]
        assert all(cands[g[0]]["importance"] >= np.float32(0.3) for g in got)


def oracle_weights(oracle):
    w = oracle.Weights()
    oracle.lib().so_weights_default(C.byref(w))
    return w


def test_tag_score_lowercases_like_str_to_lowercase(L, oracle):
    """relevance.rs:685-689 lower-cases context and tags with str::to_lowercase (full Unicode mappings, Final_Sigma) and splits
    at Unicode White_Space. Library and oracle against a restatement on top of Python's own str.lower() (same Unicode rules;
    database 13.0 here), on text in several scripts, and the generated tables are current."""
    import subprocess
    import sys
    import shodh_memory_amd as M
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    assert subprocess.run([sys.executable, os.path.join(root, "tools", "gen_unicode_lower.py"), "--check"]).returncode == 0
    WS = set([0x9, 0xA, 0xB, 0xC, 0xD, 0x20, 0x85, 0xA0, 0x1680, 0x2028, 0x2029, 0x202F, 0x205F, 0x3000] + list(range(0x2000, 0x200B)))

    def words(s):
        out, cur = [], ""
        for ch in s:
            if ord(ch) in WS:
                if cur:
                    out.append(cur)
                cur = ""
            else:
                cur += ch
        return out + ([cur] if cur else [])

    def restated(context, tags):
        if not tags:
            return np.float32(0)
        c, m = context.lower(), 0
        for tag in tags:
            t = tag.lower()
            if t in c:
                m += 1
            elif any(w.startswith(t) or t.startswith(w) for w in words(c)):
                m += 1
        return np.float32(np.float32(m) / np.float32(len(tags)))

    cases = [("Besuch in MÜNCHEN und Köln", ["münchen", "KÖLN", "berlin"]), ("ΟΔΟΣ ΑΘΗΝΑΣ", ["οδός", "οδος", "αθηνας", "αθηνασ"]),
             ("ΣΑΣ ΣΟΦΟΣ. Σ", ["σας", "σοφος", "σοφοσ", "σ"]), ("İSTANBUL ve IĞDIR", ["i̇stanbul", "istanbul", "iğdir"]),
             ("МОСКВА — столица", ["москва", "СТОЛИЦА"]), ("straße STRASSE", ["STRAẞE", "strasse"]), ("ᲐᲚᲐᲜᲘ ႠႡ", ["ალანი", "ⴀⴁ"]),
             ("\U00010400\U00010401 deseret", ["\U00010428\U00010429"]), ("tab separated　words here", ["sep", "WORDS", "her", "nbsp"]),
             ("à la carte", ["À", "car", "lac"]), ("ǅ ǈ ǋ titlecase", ["ǆ", "ǉ"]), ("Ω ohm K kelvin Å", ["ω", "k", "å"]), ("", ["x"]), ("x", [""])]
    for ctx, tags in cases:
        exp = restated(ctx, tags)
        assert np.float32(M.calculate_tag_score(ctx, tags)) == exp, (ctx, tags, M.calculate_tag_score(ctx, tags), exp)
        assert np.float32(oracle.calculate_tag_score(ctx, tags)) == exp, (ctx, tags)
    rng = np.random.default_rng(3)
    pool = [chr(c) for c in list(range(0x41, 0x5B)) + list(range(0xC0, 0x17F)) + list(range(0x386, 0x3D0)) + list(range(0x400, 0x460)) + [0x20, 0xA0, 0x2D, 0x27, 0x301, 0xB7]]
    for _ in range(300):
        ctx = "".join(rng.choice(pool, int(rng.integers(0, 30))))
        tags = ["".join(rng.choice(pool, int(rng.integers(0, 4)))) for _ in range(int(rng.integers(1, 4)))]
        if "\x00" in ctx or any("\x00" in t for t in tags):
            continue
        exp = restated(ctx, tags)
        assert np.float32(M.calculate_tag_score(ctx, tags)) == exp and np.float32(oracle.calculate_tag_score(ctx, tags)) == exp, (ctx, tags)


def test_reference_estimate_tokens_tests():
    """token_estimation.rs:93-222, the fallback of Embedder::count_tokens (minilm.rs:1216-1229), against the Python mirror"""
    from shodh_memory_amd.embedder import estimate_tokens
    assert estimate_tokens("") == 0 and estimate_tokens("hello") == 2
    text = "The quick brown fox jumps over the lazy dog. This is a typical English sentence with normal punctuation."
    assert abs(estimate_tokens(text) - len(text) // 4) <= 2
    code = '\nfn main() {\n    let mut v: Vec<String> = Vec::new();\n    for i in 0..100 {\n        v.push(format!("{}: {}", i, i * 2));\n    }\n    println!("{:?}", &v[0..5]);\n}\n'
    assert estimate_tokens(code) != len(code) // 4 and estimate_tokens(code) == len(code) * 10 // 32
    cjk = "你好世界这是一个测试用来验证中文内容的分词估算是否准确"
    assert estimate_tokens(cjk) == (len(cjk) * 3 + 1) // 2
    mixed = "This is an English sentence with a few Chinese characters 你好 in it."
    assert abs(estimate_tokens(mixed) - len(mixed.encode()) // 4) <= 3
    js = '{"users": [{"name": "Alice", "age": 30}, {"name": "Bob", "age": 25}], "total": 2}'
    assert estimate_tokens(js) == len(js) * 10 // 32
    big = "The quick brown fox jumps over the lazy dog. " * 200
    assert abs(estimate_tokens(big) - len(big) // 4) <= 2
    assert estimate_tokens("   \n\t  \n  ") > 0 and estimate_tokens("a") == 1


def test_finalize_pooled_both_branches(oracle):
    """minilm.rs:846-878: MiniLM branch (scrub + L2) and the nomic branch (parameter-free LayerNorm over the full width, Matryoshka
    truncation, L2 over the kept prefix); product vs the C restatement bit for bit, and vs an independent numpy float32 form"""
    from shodh_memory_amd import embedder as E
    rng = np.random.default_rng(9)
    for n, out_dim, pre in ((384, None, False), (768, 768, True), (768, 256, True), (768, 128, False), (16, 64, True)):
        v = (rng.standard_normal(n) * 3 + 0.7).astype(np.float32)
        v[3] = np.nan; v[5] = np.inf; v[6] = -np.inf
        got = E.finalize_pooled(v, apply_prenorm=pre, dimension=out_dim)
        exp = oracle.finalize_pooled(v, apply_prenorm=pre, out_dim=out_dim)
        assert got.tobytes() == exp.tobytes()
        x = np.where(np.isfinite(v), v, np.float32(0)).astype(np.float32)
        if pre:
            mean = np.float32(x.astype(np.float64).sum() / n)
            x = ((x - mean) / np.sqrt(np.float32(((x - mean) ** 2).astype(np.float64).sum() / n) + np.float32(1e-5))).astype(np.float32)
        x = x[:min(n, out_dim or n)]
        x = x / np.linalg.norm(x)
        assert np.abs(got - x).max() < 2e-6 and abs(np.linalg.norm(got) - 1) < 1e-6 and len(got) == min(n, out_dim or n)
    z = E.finalize_pooled(np.zeros(8, np.float32), apply_prenorm=True)          # constant input: var 0, denom = sqrt(1e-5) > EPSILON -> zeros stay zeros; norm 0 -> unchanged
    assert not z.any()


def test_environment_switches(monkeypatch):
    """SURVEY.md section 5 hooks: SHODH_TEXT_DIM, SHODH_HIP_DEVICES, SHODH_VECTOR_EXACT / SHODH_HIP_GRAPH_WALK (parsing only; no GPU)"""
    from shodh_memory_amd import index as I, _lib as L
    for k in ("SHODH_TEXT_DIM", "SHODH_HIP_DEVICES", "SHODH_VECTOR_EXACT", "SHODH_HIP_GRAPH_WALK"):
        monkeypatch.delenv(k, raising=False)
    c = I.vamana_config_from_env()
    assert (c.dimension, c.device, c.scan_mode) == (384, 0, L.SCAN_AUTO)
    monkeypatch.setenv("SHODH_TEXT_DIM", "768"); monkeypatch.setenv("SHODH_HIP_DEVICES", "3, 1"); monkeypatch.setenv("SHODH_HIP_GRAPH_WALK", "1")
    c = I.vamana_config_from_env(max_degree=16)
    assert (c.dimension, c.device, c.scan_mode, c.max_degree) == (768, 3, L.SCAN_GRAPH, 16) and I.env_devices() == [3, 1]
    monkeypatch.setenv("SHODH_VECTOR_EXACT", "1")                   # the reference's switch wins
    assert I.vamana_config_from_env().scan_mode == L.SCAN_AUTO
    monkeypatch.setenv("SHODH_TEXT_DIM", "333")
    assert I.env_dimension() == 384
