"""Host-side check of what the fp16 pre-scan's exactness rests on, with the oracle as the reference score:

1. |s~ - dot_ref| <= eps for the scores the single-query scan computes (f32 query x fp16(256 row) / 256, f32 accumulation) and for the
   MFMA scan's (query rounded to fp16 too), with the eps formula of `eps_coefficients` (scan_mfma.hip);
2. LOCAL thresholds (k <= 32): the union over slices of {s~ >= kth_local(s~) - 2 eps} contains every row of the reference top-k;
3. GLOBAL threshold (k > 32): the k-th best of per-wave maxima (scores of distinct rows) is <= the true k-th best s~, and
   {s~ >= that - 2 eps} contains every row of the reference top-k -- whatever the partition into waves is.

Pure numpy + the oracle's reference-order dot: no GPU, no product code."""
import numpy as np
import pytest

from tests import synth

f32 = np.float32


def eps_of(dim, qn, maxnorm, maxres, qres):
    """eps_coefficients + threshold_kernel / solo_scan_kernel (scalar-4 and AVX2 orders). Round 6: the two fp16 rounding terms are MEASURED -- maxres, the
    largest |row - fp16 copy| of the corpus (convert_rows_kernel), and qres, the query's own (convert_queries_kernel; 0 for the single-query scan, which keeps
    the query in f32) -- instead of charged at 2^-11 per element."""
    rel = f32(dim) * f32(1.1921e-7) * f32(1.01) + max(f32(1.0e-5), f32(dim) * f32(5.9605e-8) * f32(1.01))
    abs_a = f32(2.3842e-7) * np.sqrt(f32(dim)) * f32(1.01)
    return float((rel * maxnorm + maxres) * qn + qres * (maxnorm + maxres) + abs_a * (qn + maxnorm) + f32(1e-9))


def residual(x):
    """|x - fp16(256 x) / 256| per row, in f32 like the kernels, with their safety factor"""
    x = np.atleast_2d(x).astype(f32)
    h = (x * f32(256)).astype(np.float16).astype(np.float32) / f32(256)
    d = x - h
    return np.sqrt((d * d).sum(axis=1, dtype=np.float32)) * f32(1.0001)


def shadow_scores(rows, q, round_query):
    h = (rows * f32(256)).astype(np.float16).astype(np.float32)
    if round_query:
        qq = (q * f32(256)).astype(np.float16).astype(np.float32)
        return (h @ qq) / f32(65536.0)
    return (h @ q) / f32(256.0)


@pytest.mark.parametrize("order", [0, 1])
@pytest.mark.parametrize("dim", [128, 384, 1024])
def test_prescan_error_bound_and_windows(oracle, dim, order):
    n = 6000
    q = synth.queries(3, dim, seed=900 + dim)
    rows = synth.corpus(n, dim, seed=901 + dim, queries=q)
    rows[::7] *= f32(0.37)                                     # unnormalised rows too: the bound scales with the largest norm
    maxnorm = float(np.sqrt((rows.astype(np.float64) ** 2).sum(axis=1).max()) * 1.00001)
    maxres = float(residual(rows).max())
    assert maxres < 0.75 * 2.0 ** -11 * maxnorm            # measured, it is 0.5 - 0.65 of what the worst case charges
    ref_dot = lambda a, b: oracle.dot(a, b, order)      # noqa: E731  (reference-order dot, distance_inline.rs:67-173)
    rng = np.random.default_rng(dim)
    for qi in range(3):
        qn = float(np.sqrt((q[qi].astype(np.float64) ** 2).sum()) * 1.00001)
        ref = np.array([ref_dot(q[qi], rows[j]) for j in range(n)], f32)
        for round_query in (False, True):
            eps = eps_of(dim, qn, maxnorm, maxres, float(residual(q[qi])[0]) if round_query else 0.0)
            s = shadow_scores(rows, q[qi], round_query)
            assert np.abs(s.astype(np.float64) - ref.astype(np.float64)).max() <= eps, (dim, order, qi, round_query)
        eps = eps_of(dim, qn, maxnorm, maxres, 0.0)            # the single-query scan's (windows below)
        s = shadow_scores(rows, q[qi], False)
        for k in (1, 10, 32, 120):
            e_ids, _ = oracle.brute_force_search(rows, q[qi], k, order=order, select=True)
            need = set(e_ids.tolist())
            margin = 2.001 * eps + 1e-7
            # local thresholds: slices of uneven length
            cuts = np.sort(rng.choice(np.arange(1, n), 13, replace=False))
            got = set()
            for sl in np.split(np.arange(n), cuts):
                loc = s[sl]
                kth = np.sort(loc)[::-1][k - 1] if len(loc) >= k else -np.inf
                got |= set(sl[loc >= kth - margin - 1e-7 * abs(kth)].tolist())
            assert need <= got, (dim, order, qi, k, "local")
            # global threshold from the maxima of a random partition into "waves"
            perm = rng.permutation(n)
            parts = np.array_split(perm, 64)
            maxima = np.sort(np.array([s[p].max() for p in parts]))[::-1]
            true_kth = np.sort(s)[::-1][k - 1]
            if k <= len(maxima):
                lb = maxima[k - 1]
                assert lb <= true_kth
                got = set(np.nonzero(s >= lb - margin - 1e-7 * abs(lb))[0].tolist())
                assert need <= got, (dim, order, qi, k, "global")


@pytest.mark.parametrize("dim", [128, 384])
def test_measured_residual_bound_on_aligned_rounding_errors(oracle, dim):
    """The measured-residual bound (round 6) where Cauchy-Schwarz is tight: rows whose every element sits just short of an fp16 rounding midpoint (the
    largest residual the format allows, all of one sign) against queries parallel to that residual vector and against queries with the same property."""
    rng = np.random.default_rng(7 + dim)
    n = 400
    base = rng.standard_normal((n, dim)).astype(f32)
    base /= np.linalg.norm(base, axis=1, keepdims=True)
    h = (base * f32(256)).astype(np.float16)
    up = np.nextafter(h, np.float16(np.inf))
    mid = (h.astype(np.float64) + up.astype(np.float64)) / 2.0
    rows = (np.nextafter((mid / 256.0).astype(f32), f32(-np.inf))).astype(f32)      # rounds DOWN to h: residual ~ +half an ulp in every element
    maxnorm = float(np.sqrt((rows.astype(np.float64) ** 2).sum(axis=1).max()) * 1.00001)
    maxres = float(residual(rows).max())
    assert maxres > 0.45 * 2.0 ** -11 * maxnorm                                      # (these rows do sit near the worst case)
    res_dir = rows - (rows * f32(256)).astype(np.float16).astype(f32) / f32(256)
    queries = [res_dir[3] / np.linalg.norm(res_dir[3]), rows[5] / np.linalg.norm(rows[5]), -rows[9], res_dir[11] / np.linalg.norm(res_dir[11]) + rows[11]]
    for order in (0, 1):
        for q in queries:
            q = np.ascontiguousarray(q.astype(f32))
            qn = float(np.sqrt((q.astype(np.float64) ** 2).sum()) * 1.00001)
            ref = np.array([oracle.dot(q, rows[j], order) for j in range(n)], np.float64)
            for round_query in (False, True):
                eps = eps_of(dim, qn, maxnorm, maxres, float(residual(q)[0]) if round_query else 0.0)
                s = shadow_scores(rows, q, round_query).astype(np.float64)
                assert np.abs(s - ref).max() <= eps, (dim, order, round_query, np.abs(s - ref).max(), eps)
