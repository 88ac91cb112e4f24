"""GPU parity tests added in round 2 (all through the C ABI):
 - `shodh_fuse_scores_full_batch` (the HIP kernel of LearnedWeights::fuse_scores_full, relevance.rs:529-606) against the oracle,
   tolerance 1e-6 absolute (libm expf/log2f differ from the device's by <= 1 ulp; scores are in [0, 1]);
 - `shodh_top_k_similar` (similarity.rs:27-48): the reference's own tests + random inputs with ties against the oracle, bit-exact;
 - BASELINE.json configs[2] chained: synthetic token batches -> HIP MiniLM (bf16) -> add_vectors -> search_batch of re-encoded
   texts; ids + distances bit-equal to the oracle's brute force run on the DEVICE-PRODUCED embeddings.
"""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
FUSE_TOL = 1e-6


@pytest.fixture(scope="module")
def S():
    from shodh_memory_amd import build
    build.build()
    import shodh_memory_amd as s
    return s


def _signal_arrays(n, seed):
    rng = np.random.default_rng(seed)
    sem, ent, tag, imp, gs = (rng.random(n, dtype=np.float32) * np.float32(1.4) - np.float32(0.2) for _ in range(5))   # a little outside [0,1] too
    mom = rng.random(n, dtype=np.float32) * np.float32(2.4) - np.float32(1.2)
    acc = rng.integers(0, 5000, n).astype(np.uint32)
    acc[rng.random(n) < 0.3] = 0
    # special values: NaN / +-Inf / +-0 / subnormal / huge, in every signal (relevance.rs:601-606 guards, :1965-2066 tests)
    special = np.array([np.nan, np.inf, -np.inf, 0.0, -0.0, 1e-42, -1e-42, 3.4e38, -3.4e38, 0.5, 1.0, -1.0], np.float32)
    for arr in (sem, ent, tag, imp, mom, gs):
        pos = rng.choice(n, len(special) * 4, replace=False)
        arr[pos] = np.tile(special, 4)
    acc[rng.choice(n, 8, replace=False)] = np.array([0, 1, 2, 14, 15, 16, 0xFFFFFFFF, 0x7FFFFFFF], np.uint32)
    return sem, ent, tag, imp, mom, acc, gs


@pytest.mark.parametrize("weights", ["default", "feedback"])
def test_fuse_scores_full_batch_kernel_matches_oracle(S, oracle, weights):
    n = 20000
    sem, ent, tag, imp, mom, acc, gs = _signal_arrays(n, 7 if weights == "default" else 8)
    w = S.LearnedWeights.default()
    ow = oracle.weights_default()
    if weights == "feedback":                      # weights that moved (apply_feedback + normalize, relevance.rs:427-465)
        for args in ((1, 0, 1, 1), (0, 1, 0, 0), (0, 0, 0, 1), (1, 1, 1, 0)):
            w.apply_feedback(*args)
            oracle.weights_apply_feedback(ow, *args)
        assert np.allclose(w.as_tuple(), [ow.semantic, ow.entity, ow.tag, ow.importance, ow.momentum, ow.access_count, ow.graph_strength], atol=1e-7)
    got = w.fuse_scores_full_batch(sem, ent, tag, imp, mom, acc, gs)
    exp = np.array([oracle.fuse_scores_full(ow, float(sem[i]), float(ent[i]), float(tag[i]), float(imp[i]), float(mom[i]), int(acc[i]), float(gs[i]))
                    for i in range(n)], np.float32)
    assert np.isfinite(got).all() and np.isfinite(exp).all()           # non-finite signals calibrate to 0, never propagate
    err = np.abs(got.astype(np.float64) - exp.astype(np.float64))
    assert err.max() <= FUSE_TOL, (err.max(), int(err.argmax()))
    # the host scalar entry point and the device kernel are the same expression tree
    for i in (0, 1, 17, n - 1):
        assert abs(w.fuse_scores_full(float(sem[i]), float(ent[i]), float(tag[i]), float(imp[i]), float(mom[i]), int(acc[i]), float(gs[i])) - got[i]) <= FUSE_TOL
    assert (got >= -1e-6).all() and (got <= 1.0 + 1e-5).all()


def test_top_k_similar_reference_tests(S):
    # similarity.rs:93-122, literals restated
    top2 = S.top_k_similar([1.0, 0.0], [([1.0, 0.0], "perfect"), ([0.7, 0.7], "diagonal"), ([0.0, 1.0], "orthogonal"), ([-1.0, 0.0], "opposite")], 2)
    assert len(top2) == 2 and top2[0][1] == "perfect" and top2[1][1] == "diagonal" and top2[0][0] >= top2[1][0]
    assert len(S.top_k_similar([1.0, 0.0], [([1.0, 0.0], 1), ([0.0, 1.0], 2)], 10)) == 2
    assert S.top_k_similar([1.0, 0.0], [], 3) == []
    # cosine_similarity edge cases (:71-82): mismatched length and zero vectors give 0.0
    assert S.cosine_similarity([1.0, 2.0], [1.0, 2.0, 3.0]) == 0.0
    assert S.cosine_similarity([0.0, 0.0, 0.0], [1.0, 2.0, 3.0]) == 0.0 and S.cosine_similarity([1.0, 2.0, 3.0], [0.0, 0.0, 0.0]) == 0.0
    assert abs(S.cosine_similarity([1.0, -1.0], [-1.0, 1.0]) + 1.0) < 1e-3
    # a query of another length scores 0.0 against everything; the stable sort then keeps the input order
    r = S.top_k_similar([1.0, 0.0, 0.0], [([1.0, 0.0], "a"), ([0.0, 1.0], "b")], 2)
    assert r == [(0.0, "a"), (0.0, "b")]


@pytest.mark.parametrize("order", [0, 1])
def test_top_k_similar_matches_oracle_with_ties(S, oracle, order):
    rng = np.random.default_rng(31 + order)
    for n, dim, k in ((1, 3, 1), (37, 384, 5), (500, 384, 500), (500, 383, 20), (64, 8, 100)):
        c = rng.standard_normal((n, dim), dtype=np.float32)
        if n > 10:
            c[5] = c[2]; c[9] = c[2]; c[7] = 0.0             # exact ties (stable order decides) and a zero vector
            c[3] = -c[2]
        q = c[2].copy() if n > 2 else rng.standard_normal(dim, dtype=np.float32)
        got = S.top_k_similar(q, [(c[i], i) for i in range(n)], k, order=order)
        e_sc, e_ix = oracle.top_k_similar(q, c, k, order=order)
        assert [g[1] for g in got] == [int(i) for i in e_ix]
        assert np.array([g[0] for g in got], np.float32).tobytes() == e_sc.tobytes()


def _synth_tokens(n, max_len, seed):
    """SURVEY 8d token inputs: lengths ~U[8,128], ids ~U[1000,30521], [CLS]=101 ... [SEP]=102, right-padded to max_len"""
    g = torch.Generator(device="cuda").manual_seed(seed)
    lens = torch.randint(8, 129, (n,), generator=g, device="cuda")
    ids = torch.randint(1000, 30522, (n, max_len), generator=g, device="cuda", dtype=torch.int32)
    mask = torch.arange(max_len, device="cuda")[None, :] < lens[:, None]
    ids = torch.where(mask, ids, torch.zeros_like(ids))
    ids[:, 0] = 101
    ids[torch.arange(n, device="cuda"), lens - 1] = 102
    return ids.contiguous(), mask.to(torch.uint8).contiguous()


def test_configs2_chained_encode_add_recall(S, oracle):
    """configs[2] shape (benches/pipeline_benchmarks.rs:304-396): texts -> embed -> insert -> recall, all on the device.
    50k synthetic token sequences through the bf16 HIP encoder, appended in batches, then 64 of the SAME texts re-encoded
    as queries: self must be the top hit, and ids + distances must be bit-equal to the oracle's brute force over the
    embeddings the device produced (extract_all_vectors returns them bit-for-bit)."""
    n_texts, bsz, nq, k = 50_000, 4096, 64, 10
    e = S.MiniLMEmbedder(synthetic_seed=1234, dtype=1)
    idx = S.VamanaIndex(S.VamanaConfig(dimension=384, reserve_rows=n_texts))
    ids, mask = _synth_tokens(n_texts, 256, seed=11)
    emb = torch.empty((bsz, 384), dtype=torch.float32, device="cuda")
    first_ids = []
    for b0 in range(0, n_texts, bsz):
        b = min(bsz, n_texts - b0)
        e.encode_ids_device(ids[b0:b0 + b].contiguous(), mask[b0:b0 + b].contiguous(), out=emb[:b])
        torch.cuda.synchronize()
        first_ids.append(idx.add_vectors(emb[:b]))
    assert first_ids == list(range(0, n_texts, bsz)) and idx.len() == n_texts       # ids dense, in insertion order (vamana.rs:854-855)
    rows = idx.extract_all_vectors()
    norms = np.linalg.norm(rows, axis=1)
    assert np.abs(norms - 1).max() < 1e-3                                            # unit vectors out of finalize_pooled
    pick = np.linspace(0, n_texts - 1, nq).astype(np.int64)
    qemb = torch.empty((nq, 384), dtype=torch.float32, device="cuda")
    sel = torch.from_numpy(pick).cuda()
    e.encode_ids_device(ids[sel].contiguous(), mask[sel].contiguous(), out=qemb)
    torch.cuda.synchronize()
    q = qemb.cpu().numpy()
    # the encoder is deterministic per sequence, whatever batch it is encoded in (packed tokens, no cross-sequence op)
    assert np.abs(q - rows[pick]).max() < 2e-3
    got_ids, got_dist, counts = idx.search_batch(q, k)
    stats = idx.scan_stats()
    exp_ids, exp_dist = oracle.brute_force_batch(rows, q, k)
    assert (counts == k).all()
    assert np.array_equal(got_ids, exp_ids), "top-k ids differ from the oracle on device-produced embeddings"
    assert got_dist.tobytes() == exp_dist.tobytes()
    assert (got_ids[:, 0] == pick).all() or (np.abs(got_dist[:, 0] + 1) < 2e-3).all()  # self (or an exact duplicate text) is the top hit
    assert (got_dist[:, 0] < -0.99).all()
    # recall through the device-pointer API agrees with the host-pointer one
    d_ids, d_dist, d_cnt = idx.search_batch_device(qemb, k)
    torch.cuda.synchronize()
    assert np.array_equal(d_ids.cpu().numpy().view(np.uint32), got_ids) and d_dist.cpu().numpy().tobytes() == got_dist.tobytes()
    print("configs[2] chained: %d texts, %d queries; pre-scan emitted %d, re-scored %d, exact-fallback queries %d"
          % (n_texts, nq, stats["emitted"], stats["rescored"], stats["overflowed"]))
