"""The host-glue rows against evidence that does NOT come from oracle/: the library's `shodh_search_ids_postprocess` and
`shodh_rrf_fuse` are C++ retypings of the oracle's loops ("checked against a twin"), so here they meet

  * the reference's own unit tests, restated literally with fixed ids (src/memory/hybrid_search.rs:969-1056, :1171-1240);
  * a third implementation written the way the Rust is written -- a HashMap keyed by MemoryId, `and_modify / or_insert`, `sort_by(total_cmp desc,
    then id asc)`, `truncate` (src/memory/retrieval.rs:927-961, hybrid_search.rs:556-594) -- in Python, numpy.float32 arithmetic, on random inputs;
  * Python's `str.lower()` over every Unicode code point and around capital sigma for the lower-casing table behind `calculate_tag_score`
    (relevance.rs:685-689) -- the table in csrc/ and the one in oracle/ are the same generated file, so the oracle cannot catch an error in it.

Nothing here imports oracle/."""
import ctypes as C
import functools
import struct
import sys
import unicodedata
import uuid

import numpy as np
import pytest

f32 = np.float32


@pytest.fixture(scope="module")
def lib():
    from shodh_memory_amd import build
    build.build()
    from shodh_memory_amd import _lib
    return _lib.lib()


# ---- f32::total_cmp, Uuid ordering ------------------------------------------------------------------------------------
def total_cmp(a, b):
    """f32::total_cmp: sign-magnitude bits as a two's-complement-comparable integer (library/core/src/num/f32.rs)"""
    def key(x):
        i = struct.unpack("<i", struct.pack("<f", float(x)))[0]
        return i ^ (((i >> 31) & 0xFFFFFFFF) >> 1)
    ka, kb = key(a), key(b)
    return (ka > kb) - (ka < kb)


def by_score_desc_then_id(a, b):
    """|a, b| b.1.total_cmp(&a.1).then_with(|| a.0.cmp(&b.0)) -- retrieval.rs:958, hybrid_search.rs:591"""
    c = total_cmp(b[1], a[1])
    if c:
        return c
    return (a[0] > b[0]) - (a[0] < b[0])             # Uuid: Ord over its 16 bytes


# ---- RRFusion (hybrid_search.rs:529-594), written like the Rust ----------------------------------------------------------
def rrf_new(k, weights):
    s = f32(0.0)
    for w in weights:
        s = f32(s + f32(w))
    if s > 0:
        norm = [f32(f32(w) / s) for w in weights]
    else:
        norm = [f32(f32(1.0) / f32(len(weights)))] * len(weights)
    return f32(k), norm


def rrf_fuse_py(k, weights, ranked_lists):
    k, norm = rrf_new(k, weights)
    scores = {}
    for li, lst in enumerate(ranked_lists):
        w = norm[li] if li < len(norm) else f32(1.0)
        for rank, mid in enumerate(lst):
            contrib = f32(w / f32(k + f32(rank + 1)))
            scores[mid] = f32(scores.get(mid, f32(0.0)) + contrib)
    out = sorted(scores.items(), key=functools.cmp_to_key(by_score_desc_then_id))
    return [m for m, _ in out], np.array([s for _, s in out], f32)


def rrf_fuse_lib(lib, k, weights, lists):
    w = np.asarray(weights, f32)
    flat = np.frombuffer(b"".join(b"".join(l) for l in lists) or b"\0" * 16, np.uint8).copy()
    lens = (C.c_size_t * max(len(lists), 1))(*[len(l) for l in lists])
    cap = max(sum(len(l) for l in lists), 1)
    ou = np.zeros((cap, 16), np.uint8)
    os_ = np.zeros(cap, f32)
    m = lib.shodh_rrf_fuse(k, w.ctypes.data, len(lists), flat.ctypes.data, lens, ou.ctypes.data, os_.ctypes.data, cap)
    return [bytes(ou[i]) for i in range(m)], os_[:m].copy()


def U(i):
    return uuid.UUID(int=i).bytes


def test_rrf_reference_unit_tests_literally(lib):
    # test_rrf_fusion_basic (hybrid_search.rs:969-1011)
    id1, id2, id3 = U(0x1111), U(0x2222), U(0x3333)
    ids, sc = rrf_fuse_lib(lib, 60.0, [0.5, 0.5], [[id1, id2, id3], [id2, id1, id3]])
    s = dict(zip(ids, sc))
    assert len(ids) == 3 and abs(s[id1] - s[id2]) < 1e-4 and s[id3] < s[id1] and s[id3] < s[id2] and ids[2] == id3
    # expected values by hand: 0.5/61 + 0.5/62 and 2 * 0.5/63, in f32 like the Rust
    assert s[id1] == f32(f32(f32(0.5) / f32(61.0)) + f32(f32(0.5) / f32(62.0))) and s[id3] == f32(f32(f32(0.5) / f32(63.0)) + f32(f32(0.5) / f32(63.0)))
    assert s[id2] == f32(f32(f32(0.5) / f32(62.0)) + f32(f32(0.5) / f32(61.0)))           # list order of the additions: list1 first
    # test_rrf_fusion_disjoint (:1013-1029)
    ids, sc = rrf_fuse_lib(lib, 60.0, [0.5, 0.5], [[id1], [id2]])
    assert len(ids) == 2 and abs(sc[0] - sc[1]) < 0.001
    # test_rrf_fusion_ties_break_by_memory_id (:1031-1056): fixed-byte ids, the HIGHER one first in the input
    lo, hi = bytes([0x00] * 16), bytes([0xFF] * 16)
    ids, sc = rrf_fuse_lib(lib, 60.0, [0.5, 0.5], [[hi], [lo]])
    assert len(ids) == 2 and abs(sc[0] - sc[1]) < np.finfo(f32).eps and ids == [lo, hi]
    # test_rrf_weighted_fusion (:1171-1195)
    bm25, vec = [id1, id2], [id2, id1]
    assert rrf_fuse_lib(lib, 60.0, [0.8, 0.2], [bm25, vec])[0][0] == id1, "BM25-heavy should favor BM25 winner"
    assert rrf_fuse_lib(lib, 60.0, [0.2, 0.8], [bm25, vec])[0][0] == id2, "Vector-heavy should favor vector winner"
    # test_rrf_k_parameter_effect (:1197-1240)
    l1, l2 = [id1, id2, id3], [id3, id2, id1]
    lo_ids, lo_sc = rrf_fuse_lib(lib, 1.0, [0.5, 0.5], [l1, l2])
    hi_ids, hi_sc = rrf_fuse_lib(lib, 100.0, [0.5, 0.5], [l1, l2])
    rel_lo = dict(zip(lo_ids, lo_sc))[id2] / lo_sc[0]
    rel_hi = dict(zip(hi_ids, hi_sc))[id2] / hi_sc[0]
    assert rel_hi >= rel_lo - 0.01, "High k should be more forgiving of rank variation"
    # RRFusion::new with weights summing to zero: uniform (:541-545); more lists than weights: weight 1.0 (:561)
    ids, sc = rrf_fuse_lib(lib, 60.0, [0.0, 0.0], [[hi], [lo]])
    assert ids == [lo, hi] and sc[0] == f32(f32(0.5) / f32(61.0))


def test_rrf_equals_the_hashmap_restatement_on_random_lists(lib):
    rng = np.random.default_rng(11)
    pool = [U(int(x)) for x in rng.integers(1, 2 ** 62, 60)]
    for trial in range(60):
        n_lists = int(rng.integers(1, 5))
        lists = [[pool[int(i)] for i in rng.permutation(60)[:int(rng.integers(0, 40))]] for _ in range(n_lists)]
        if trial % 3 == 0:                                  # the constant collisions the reference's comment talks about: same ranks in several lists
            lists = [list(l) for l in lists]
            for l in lists[1:]:
                l[:5] = lists[0][:5][::-1][:len(l[:5])]
            lists = [list(dict.fromkeys(l)) for l in lists]
        w = (rng.random(n_lists) * (0 if trial == 7 else 1)).astype(f32)
        k = float(rng.choice([1.0, 45.0, 60.0, 100.0]))
        a_ids, a_sc = rrf_fuse_lib(lib, k, w, lists)
        e_ids, e_sc = rrf_fuse_py(k, w, lists)
        assert a_ids == e_ids, trial
        assert a_sc.tobytes() == e_sc.tobytes(), trial


# ---- search_ids post-processing (retrieval.rs:927-961), written like the Rust ---------------------------------------------------
def search_ids_py(results, id_mapping, limit):
    best = {}
    for vector_id, distance in results:
        similarity = f32(-f32(distance))
        mem = id_mapping.get(int(vector_id))
        if mem is None:
            continue
        if mem in best:
            if similarity > best[mem]:                     # and_modify: a plain `>` (not total_cmp): -0.0 does not replace 0.0
                best[mem] = similarity
        else:
            best[mem] = similarity
    out = sorted(best.items(), key=functools.cmp_to_key(by_score_desc_then_id))[:limit]
    return [m for m, _ in out], np.array([s for _, s in out], f32)


def search_ids_lib(lib, vec_ids, dists, v2m, limit):
    vec_ids = np.ascontiguousarray(vec_ids, np.uint32)
    dists = np.ascontiguousarray(dists, f32)
    ou = np.zeros((max(limit, 1), 16), np.uint8)
    os_ = np.zeros(max(limit, 1), f32)
    m = lib.shodh_search_ids_postprocess(vec_ids.ctypes.data, dists.ctypes.data, len(vec_ids), v2m.ctypes.data, len(v2m), limit, ou.ctypes.data, os_.ctypes.data)
    return [bytes(ou[i]) for i in range(m)], os_[:m].copy()


def mapping_table(n_vec, id_mapping):
    """the library's form of IdMapping::get_memory_id: [n_vec][16], all-0xFF = no memory for this vector id"""
    t = np.full((n_vec, 16), 0xFF, np.uint8)
    for v, m in id_mapping.items():
        t[v] = np.frombuffer(m, np.uint8)
    return t


def test_search_ids_literal_cases(lib):
    a, b, c = U(0xA), U(0xB), U(0xC)
    idm = {0: a, 1: a, 2: b, 3: c, 4: c, 5: b}
    v2m = mapping_table(8, idm)                              # vector ids 6, 7: no memory (deleted / never mapped): skipped
    # chunks of one memory: the best chunk's similarity stands for the memory (max over chunks); ascending distance in, descending similarity out
    res = [(0, -0.9), (2, -0.8), (1, -0.7), (6, -0.65), (3, -0.6), (5, -0.5), (4, -0.4)]
    ids, sim = search_ids_lib(lib, [r[0] for r in res], [r[1] for r in res], v2m, 10)
    assert ids == [a, b, c] and sim.tolist() == [f32(0.9), f32(0.8), f32(0.6)]
    ids, sim = search_ids_lib(lib, [r[0] for r in res], [r[1] for r in res], v2m, 2)          # truncate(limit)
    assert ids == [a, b]
    # equal similarities: MemoryId ascending decides, whatever the input order
    res = [(3, -0.5), (2, -0.5), (0, -0.5)]
    ids, sim = search_ids_lib(lib, [r[0] for r in res], [r[1] for r in res], v2m, 10)
    assert ids == [a, b, c] and sim.tolist() == [f32(0.5)] * 3
    # total_cmp: +0.0 sorts above -0.0 (distance -0.0 -> similarity +0.0), and `>` does not let a later -0.0 / +0.0 replace the first chunk's value
    res = [(0, 0.0), (2, -0.0), (1, -0.0)]
    ids, sim = search_ids_lib(lib, [r[0] for r in res], [r[1] for r in res], v2m, 10)
    assert ids == [b, a] and np.signbit(sim).tolist() == [False, True]
    # vector ids beyond the table, an empty result list, limit 0
    assert search_ids_lib(lib, [100, 7], [-0.9, -0.8], v2m, 5)[0] == []
    assert search_ids_lib(lib, [], [], v2m, 5)[0] == []
    assert search_ids_lib(lib, [0], [-0.9], v2m, 0)[0] == []


def test_search_ids_equals_the_hashmap_restatement_on_random_results(lib):
    rng = np.random.default_rng(3)
    n_vec = 400
    mems = [U(int(x)) for x in rng.integers(1, 2 ** 62, 90)]
    idm = {v: mems[int(rng.integers(0, 90))] for v in range(n_vec) if rng.random() > 0.12}
    v2m = mapping_table(n_vec, idm)
    for trial in range(80):
        n = int(rng.integers(0, 250))
        vec_ids = rng.integers(0, n_vec + 30, n).astype(np.uint32)
        d = np.sort(-rng.random(n).astype(f32))                             # an index hands over ascending distances
        d[rng.random(n) < 0.25] = f32(-0.5)                                 # ties: the id order decides
        if trial % 5 == 0 and n:
            d[rng.random(n) < 0.2] = f32(0.0)
            d[rng.random(n) < 0.2] = f32(-0.0)
        limit = int(rng.integers(1, 50))
        a_ids, a_sim = search_ids_lib(lib, vec_ids, d, v2m, limit)
        e_ids, e_sim = search_ids_py(list(zip(vec_ids.tolist(), d.tolist())), idm, limit)
        assert a_ids == e_ids, trial
        assert a_sim.tobytes() == e_sim.tobytes(), trial


# ---- str::to_lowercase table against Python's str.lower() ------------------------------------------------------------------------------
def lower_lib(lib, s):
    b = s.encode("utf-8")
    out = C.create_string_buffer(4 * len(b) + 16)
    n = lib.shodh_to_lowercase(b, out, len(out))
    assert n < len(out)
    return out.raw[:n].decode("utf-8")


@pytest.mark.skipif(unicodedata.unidata_version != "13.0.0", reason="the table was generated from Unicode 13.0.0 (Python 3.10)")
def test_lowercase_table_equals_str_lower_over_every_code_point(lib):
    """every scalar value on its own, in chunks (a space between two code points keeps the Final_Sigma context out of it; U+03A3 itself is in the next test)"""
    cps = [cp for cp in range(1, 0x110000) if not (0xD800 <= cp <= 0xDFFF) and cp != 0x3A3]
    bad = []
    for i in range(0, len(cps), 4096):
        chunk = cps[i:i + 4096]
        s = " ".join(chr(cp) for cp in chunk)
        got = lower_lib(lib, s)
        want = s.lower()
        if got != want:
            for cp in chunk:                                 # name the code points
                if lower_lib(lib, chr(cp)) != chr(cp).lower():
                    bad.append(hex(cp))
    assert not bad, bad[:20]
    changed = sum(1 for cp in cps if chr(cp).lower() != chr(cp))
    assert changed >= 1390                                   # (the table's size: the check above was not vacuous)


def test_lowercase_final_sigma_and_special_cases(lib):
    cases = ["ΟΔΥΣΣΕΥΣ", "ΑΣ", "Σ", "ΣΑ", "ΑΣ.", "ΑΣ Σ", "A.Σ", "ΆΣ́", "ΑΣ­Α", "aΣ'", "'Σ", "1Σ", "ΑΣΣ", "Σ Σ", "ΣΣ", "İstanbul", "ǅ ǈ ǋ ǲ", "ẞ", "ΆΈΉ",
             "HELLO World", "Straße", "ÀÉÎÕÜ", "ԱԲԳ", "ᏣᎳᎩ", "Ⓐⓑ", "𐐀𐐁", "𞤀𞤁", "Ǆ", "K Å Ω", "ΑΣͅ", "ΑΣʰΒ", ":Σ", "Α:Σ", "Α:Σ:Α"]
    for s in cases:
        assert lower_lib(lib, s) == s.lower(), (s, lower_lib(lib, s), s.lower())
    # every code point right before and right after a capital sigma (the Final_Sigma rule looks both ways through Case_Ignorable characters)
    if unicodedata.unidata_version == "13.0.0":
        bad = []
        for cp in list(range(1, 0x3000)) + list(range(0xA640, 0xABFF)) + list(range(0xFB00, 0xFFFF)) + list(range(0x10400, 0x10500)) + list(range(0x1D400, 0x1D800)) + list(range(0xE0001, 0xE0200)):
            if 0xD800 <= cp <= 0xDFFF:
                continue
            ch = chr(cp)
            for s in (ch + "Σ", "Α" + ch + "Σ", "ΑΣ" + ch, "ΑΣ" + ch + "Α", "Σ" + ch):
                if lower_lib(lib, s) != s.lower():
                    bad.append((hex(cp), s))
        assert not bad, bad[:10]


def test_tag_score_uses_that_lowercase(lib):
    """calculate_tag_score (relevance.rs:680-705): tags and context lower-cased, substring or word-prefix match, share of the tags that match"""
    def score(ctx, tags):
        arr = (C.c_char_p * max(len(tags), 1))(*[t.encode("utf-8") for t in tags])
        lib.shodh_calculate_tag_score.restype = C.c_float
        return lib.shodh_calculate_tag_score(ctx.encode("utf-8"), arr, len(tags))
    assert score("Working on the RUST compiler", ["rust", "Compiler", "python"]) == f32(2.0 / 3.0)
    assert score("ΟΔΥΣΣΕΥΣ sails", ["οδυσσευς"]) == 1.0                      # final sigma in the context, written out in the tag
    assert score("İSTANBUL trip", ["i̇stanbul"]) == 1.0                   # U+0130 lower-cases to two code points
    assert score("anything", []) == 0.0
