"""Seeded synthetic corpus / queries: 50% correlated rows
normalize(1 + 0.5*sqrt(D)*e_{i mod D} + 0.3*N(0,I)) (the reference's own test_vector fixture,
retrieval.rs:2430-2438, plus noise so rows stay distinct past i = D), 50% i.i.d. N(0,I) unit rows,
plus an adversarial slice: 1% exact duplicates and 0.1% rows equal to a query."""
import numpy as np

SEED = 20260926


def corpus(n, dim=384, seed=SEED, queries=None, adversarial=True):
    rng = np.random.default_rng(seed)
    x = rng.standard_normal((n, dim), dtype=np.float32)
    half = n // 2
    idx = np.arange(half)
    x[:half] *= np.float32(0.3)
    x[:half] += np.float32(1.0)
    x[idx, idx % dim] += np.float32(0.5) * np.sqrt(np.float32(dim))
    x /= np.linalg.norm(x, axis=1, keepdims=True).astype(np.float32)
    perm = rng.permutation(n)
    x = np.ascontiguousarray(x[perm])
    if adversarial and n >= 200:
        nd = max(n // 100, 1)
        src = rng.integers(0, n, nd)
        dst = rng.integers(0, n, nd)
        x[dst] = x[src]
        if queries is not None and len(queries):
            nqd = max(n // 1000, 1)
            dst = rng.integers(0, n, nqd)
            x[dst] = queries[rng.integers(0, len(queries), nqd)]
    return x


def queries(nq, dim=384, seed=SEED + 1):
    return corpus(nq, dim, seed, adversarial=False)


def tombstones(n, frac=0.05, seed=SEED + 2):
    rng = np.random.default_rng(seed)
    d = np.zeros(n, np.uint8)
    d[rng.choice(n, int(n * frac), replace=False)] = 1
    return d


def clustered_corpus(n, dim=384, n_clusters=64, within=0.8, between=0.3, seed=SEED + 7):
    """Mixture of vMF-like clusters: rows of one cluster have pairwise cosine ~`within`, rows of different
    clusters ~`within * between` .. `between` -- far denser than the half-i.i.d. corpus above, the regime of real sentence
    embeddings. centre_c = normalize(sqrt(between) u + sqrt(1 - between) g_c); row = normalize(sqrt(within) centre + sqrt(1 - within) noise)."""
    rng = np.random.default_rng(seed)
    u = rng.standard_normal(dim).astype(np.float32); u /= np.linalg.norm(u)
    g = rng.standard_normal((n_clusters, dim)).astype(np.float32); g /= np.linalg.norm(g, axis=1, keepdims=True)
    cent = np.sqrt(np.float32(between)) * u[None, :] + np.sqrt(np.float32(1 - between)) * g
    cent /= np.linalg.norm(cent, axis=1, keepdims=True)
    lab = rng.integers(0, n_clusters, n)
    noise = rng.standard_normal((n, dim), dtype=np.float32)
    noise /= np.linalg.norm(noise, axis=1, keepdims=True)
    x = np.sqrt(np.float32(within)) * cent[lab] + np.sqrt(np.float32(1 - within)) * noise
    x /= np.linalg.norm(x, axis=1, keepdims=True)
    return np.ascontiguousarray(x.astype(np.float32)), lab
