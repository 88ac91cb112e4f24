"""VAMA v1 / SPAN v1 file formats: csrc/persist.hip against an independent restatement of
the layouts written with struct.pack from the reference's format docs (vamana_persist.rs:6-34, :98-112;
spann.rs:13-52, :221-252). Host-only code: runs without a GPU."""
import os
import struct

import numpy as np
import pytest

from shodh_memory_amd import _lib as L
from shodh_memory_amd import persist as P


def fnv1a64(data):                                   # vamana_persist.rs:155-163 == spann.rs:1091-1098
    h = 0xcbf29ce484222325
    for b in data:
        h ^= b
        h = (h * 0x100000001b3) & 0xFFFFFFFFFFFFFFFF
    return h


def align(o, a=64):
    return (o + a - 1) & ~(a - 1)


def ref_write_vama(vectors, max_degree, medoid, metric, deleted, incr, graph):
    """restates VamanaIndex::save_to_file (vamana_persist.rs:175-284)"""
    n, d = vectors.shape
    body = b"".join(struct.pack("<I", i) for i in deleted)
    for nb in graph:
        body += struct.pack("<H", len(nb)) + b"".join(struct.pack("<I", x) for x in nb)
    voff = align(64 + len(body))
    body += b"\0" * (voff - 64 - len(body)) + vectors.astype("<f4").tobytes()

    def header(ck):
        h = b"VAMA" + struct.pack("<IQIII", 1, n, d, max_degree, medoid) + bytes([metric]) + struct.pack("<IQQ", len(deleted), incr, ck)
        return h + b"\0" * (64 - len(h))
    return header(fnv1a64(body)) + body


def ref_write_span(num_vectors, cent, cb, lists, codes, metric):
    """restates SpannIndex::save_to_file (spann.rs:750-876); lists = [ids per partition], codes = [codes per partition] or None"""
    Pn, D = cent.shape
    pq = cb is not None
    M = D // 8 if pq else 0
    coff = align(128)
    boff = align(coff + Pn * D * 4)
    bsz = 12 + M * 256 * 8 * 4 if pq else 0
    ioff = align(boff + bsz)
    doff = align(ioff + Pn * 12)
    esz = 4 + M
    total = sum(len(l) for l in lists)
    buf = bytearray(doff + total * esz)
    buf[coff:coff + Pn * D * 4] = cent.astype("<f4").tobytes()
    if pq:
        buf[boff:boff + 12] = struct.pack("<III", M, 256, 8)
        buf[boff + 12:boff + 12 + M * 256 * 8 * 4] = cb.astype("<f4").tobytes()
    w = 0
    for p, ids in enumerate(lists):
        buf[ioff + p * 12:ioff + p * 12 + 12] = struct.pack("<QI", w, len(ids))
        o = doff + w
        for i, vid in enumerate(ids):
            buf[o:o + 4] = struct.pack("<I", vid)
            if pq:
                buf[o + 4:o + 4 + M] = bytes(codes[p][i])
            o += esz
        w += len(ids) * esz
    ck = fnv1a64(bytes(buf[128:]))
    h = b"SPAN" + struct.pack("<IQII", 1, num_vectors, Pn, D) + bytes([1 if pq else 0]) + struct.pack("<I", M) + bytes([metric]) + struct.pack("<QQQQQ", ck, coff, boff, ioff, doff)
    buf[0:len(h)] = h
    return bytes(buf)


@pytest.fixture(scope="module")
def lib():
    return L.lib()


def test_fnv_kats():
    assert fnv1a64(b"") == 0xcbf29ce484222325 and fnv1a64(b"a") == 0xaf63dc4c8601ec8c


def test_vama_header_fields(lib, tmp_path):
    # the values of the reference's test_header_serialization (vamana_persist.rs:504-533), at their byte offsets (:98-112)
    rng = np.random.default_rng(1)
    vec = rng.standard_normal((7, 4)).astype(np.float32)
    path = tmp_path / "h.vamana"
    P.write_vamana(path, vec, max_degree=32, medoid=42, metric=0, deleted=[1, 3, 4, 5, 6], incremental_inserts=100)
    raw = path.read_bytes()
    assert raw[:4] == b"VAMA" and struct.unpack_from("<I", raw, 4)[0] == 1
    assert struct.unpack_from("<Q", raw, 8)[0] == 7 and struct.unpack_from("<III", raw, 16) == (4, 32, 42)
    assert raw[28] == 0 and struct.unpack_from("<I", raw, 29)[0] == 5 and struct.unpack_from("<Q", raw, 33)[0] == 100
    assert struct.unpack_from("<Q", raw, 41)[0] == fnv1a64(raw[64:]) and raw[49:64] == b"\0" * 15
    info = P.vama_info(path)
    assert (info["num_vectors"], info["dimension"], info["max_degree"], info["medoid"], info["deleted_count"], info["incremental_inserts"]) == (7, 4, 32, 42, 5, 100)


def test_vama_reads_reference_layout_and_writes_it_back_identically(lib, tmp_path):
    rng = np.random.default_rng(2)
    n, d = 37, 24
    vec = rng.standard_normal((n, d)).astype(np.float32)
    graph = [list(rng.choice(n, size=int(rng.integers(0, 9)), replace=False).astype(int)) for _ in range(n)]
    deleted = [3, 9, 20]
    blob = ref_write_vama(vec, 8, 5, 0, deleted, 12, graph)
    path = tmp_path / "ref.vamana"
    path.write_bytes(blob)
    assert P.verify_index_file(path)
    f = P.read_vamana(path, with_graph=True)
    assert np.array_equal(f["vectors"], vec) and list(f["deleted"]) == deleted
    assert list(f["degree"]) == [len(g) for g in graph] and list(f["neighbors"]) == [x for g in graph for x in g]
    out = tmp_path / "ours.vamana"
    P.write_vamana(out, f["vectors"], max_degree=8, medoid=5, metric=0, deleted=f["deleted"], incremental_inserts=12, degree=f["degree"], neighbors=f["neighbors"])
    assert out.read_bytes() == blob                       # byte-identical round trip, checksum included


def test_vama_checksum_detects_corruption(lib, tmp_path):
    # vamana_persist.rs:473-501: flip bits at HEADER_SIZE + 10 -> verify_index_file is false, load fails
    vec = np.array([[1, 0, 0, 0], [0, 1, 0, 0]], np.float32)
    path = tmp_path / "corrupt.vamana"
    P.write_vamana(path, vec)
    raw = bytearray(path.read_bytes())
    raw[64 + 10] ^= 0xFF
    path.write_bytes(bytes(raw))
    assert not P.verify_index_file(path)
    with pytest.raises(L.ShodhError):
        P.read_vamana(path)
    bad = tmp_path / "bad.vamana"
    bad.write_bytes(b"VAMX" + bytes(raw[4:]))
    with pytest.raises(L.ShodhError):
        P.vama_info(bad)
    with pytest.raises(L.ShodhError):
        P.vama_info(tmp_path / "missing.vamana")


def test_vama_empty_adjacency_file(lib, tmp_path):
    vec = np.eye(5, 4, dtype=np.float32)
    path = tmp_path / "e.vamana"
    P.write_vamana(path, vec, max_degree=8)
    assert path.read_bytes() == ref_write_vama(vec, 8, 0, 0, [], 0, [[] for _ in range(5)])


@pytest.mark.parametrize("pq", [True, False])
def test_span_reads_reference_layout_and_writes_it_back_identically(lib, tmp_path, pq):
    rng = np.random.default_rng(3)
    Pn, D = 5, 16
    M = D // 8
    cent = rng.standard_normal((Pn, D)).astype(np.float32)
    cb = rng.standard_normal((M, 256, 8)).astype(np.float32) if pq else None
    lens = [4, 0, 7, 1, 3]
    lists, codes, nxt = [], [], 100
    for ln in lens:
        lists.append(list(range(nxt, nxt + ln)))
        codes.append(rng.integers(0, 256, size=(ln, M), dtype=np.uint8))
        nxt += ln + 5
    blob = ref_write_span(sum(lens), cent, cb, lists, codes if pq else None, 0)
    path = tmp_path / "ref.spann"
    path.write_bytes(blob)
    f = P.read_spann(path)
    info = f["info"]
    assert (info["num_vectors"], info["num_partitions"], info["dimension"], info["pq_enabled"], info["total_postings"]) == (sum(lens), Pn, D, 1 if pq else 0, sum(lens))
    assert np.array_equal(f["centroids"], cent)
    assert list(f["list_off"]) == list(np.concatenate([[0], np.cumsum(lens)]))
    assert list(f["ids"]) == [x for l in lists for x in l]
    if pq:
        assert (info["pq_subvectors"], info["pq_num_centroids"], info["pq_subvec_dim"]) == (M, 256, 8)
        assert np.array_equal(f["codebook"], cb) and np.array_equal(f["codes"], np.concatenate(codes))
    out = tmp_path / "ours.spann"
    P.write_spann(out, sum(lens), f["centroids"], f["codebook"], f["list_off"], f["ids"], f["codes"])
    assert out.read_bytes() == blob


def test_span_rejects_corruption_and_empty(lib, tmp_path):
    cent = np.ones((2, 8), np.float32)
    blob = bytearray(ref_write_span(0, cent, None, [[], []], None, 0))
    blob[200] ^= 1
    p = tmp_path / "c.spann"
    p.write_bytes(bytes(blob))
    with pytest.raises(L.ShodhError):
        P.span_info(p)
    with pytest.raises(L.ShodhError):                   # "Cannot save empty index" (spann.rs:757-759)
        P.write_spann(tmp_path / "z.spann", 0, np.zeros((0, 8), np.float32), None, np.zeros(1, np.uint64), np.zeros(0, np.uint32), None)
