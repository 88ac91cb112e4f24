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


def eps_of(dim, qn, maxnorm):
    """eps_coefficients + threshold_kernel / solo_scan_kernel (scalar-4 and AVX2 orders)"""
    rel = f32(9.7704e-4) + f32(dim) * f32(1.1921e-7) * f32(1.01) + f32(1.0e-5)
    abs_a = f32(2.3842e-7) * np.sqrt(f32(dim)) * f32(1.01)
    return float(rel * maxnorm * qn + abs_a * (qn + maxnorm) + f32(1e-9))


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
    ref_dot = lambda a, b: oracle.dot(a, b, order)      # noqa: E731  (reference-order dot, distance_inline.rs:67-173)
    rng = np.random.default_rng(dim)
    for qi in range(3):
        qn = float(np.sqrt((q[qi].astype(np.float64) ** 2).sum()) * 1.00001)
        eps = eps_of(dim, qn, maxnorm)
        ref = np.array([ref_dot(q[qi], rows[j]) for j in range(n)], f32)
        for round_query in (False, True):
            s = shadow_scores(rows, q[qi], round_query)
            assert np.abs(s.astype(np.float64) - ref.astype(np.float64)).max() <= eps, (dim, order, qi, round_query)
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
