"""Recall Layer 4 (hybrid leg + graph leg fusion, src/memory/mod.rs:3878-4468): the library's host code against the
oracle's C restatement (bit-for-bit in every mode) and against an independent numpy float32 restatement of the default
(calibrated-max, fitted symmetric gate) mode. The reference holds no literal-valued test for this block; the density weights
are checked against graph_retrieval.rs:2447-2472."""
import ctypes as C
import itertools
import uuid

import numpy as np
import pytest

f32 = np.float32


@pytest.fixture(scope="module")
def M():
    from shodh_memory_amd import build
    build.build()
    import shodh_memory_amd as m
    return m


def _case(rng, n_mem=60):
    """A hybrid leg (ranked ids + (bm25, vector) components) and a graph leg drawn the way recall sees them."""
    mem = [uuid.UUID(int=int(x)).bytes for x in rng.integers(1, 2 ** 62, n_mem)]
    nh = int(rng.integers(0, 50))
    ng = int(rng.integers(0, 14))
    hyb = []
    for i in rng.integers(0, n_mem, nh):                       # repeats allowed: the later components win
        bm = f32(rng.gamma(2.0, 20.0)) if rng.random() < 0.7 else f32(0.0)
        ve = f32(rng.uniform(0.2, 0.8)) if rng.random() < 0.6 else f32(0.0)
        if rng.random() < 0.03:
            ve = f32(-0.1)                                        # a negative vector score clamps to 0
        hyb.append((mem[int(i)], float(bm), float(ve)))
    gr = [(mem[int(i)], float(f32(rng.uniform(0.0, 1.3)))) for i in rng.integers(0, n_mem, ng)]
    if ng and rng.random() < 0.2:
        gr[0] = (gr[0][0], gr[-1][1])                             # equal activations
    return hyb, gr, int(rng.integers(0, 200))


MODES = [dict(), dict(fusion_rrf=1), dict(fusion_v2=1), dict(fusion_sum=1), dict(fusion_flat=1, fusion_v2=1),
         dict(flat_adaptive=0), dict(adapt_feature=1), dict(adapt_feature=2), dict(adapt_symmetric=0),
         dict(adapt_feature=1, agree_k=3.0, agree_lo=0.2, agree_hi=0.9), dict(isolate_leg=1), dict(isolate_leg=2), dict(isolate_leg=3),
         dict(flat_consensus=1.7, adapt_trust_max=3.5), dict(fusion_rrf=1, isolate_leg=3), dict(graph_w=0.5, hybrid_w=0.5, rrf_k=60.0)]


def test_library_equals_oracle_in_every_mode(M, oracle):
    rng = np.random.default_rng(11)
    cases = [_case(rng) for _ in range(40)] + [([], [], 0), ([], [(b"\x01" * 16, 0.5)], 7), ([(b"\x02" * 16, 0.0, 0.0)], [], 3)]
    for mode in MODES:
        lf = M.LegFusion(**mode)
        for hyb, gr, qlen in cases:
            got, trust = lf.fuse(hyb, gr, qlen)
            eu, es, et = oracle.fuse_legs(hyb, gr, qlen, **mode)
            assert [g[0] for g in got] == eu, mode
            assert np.array([g[1] for g in got], f32).tobytes() == es.tobytes(), mode
            assert f32(trust).tobytes() == f32(et).tobytes(), mode


# ---- an independent restatement of the DEFAULT mode in numpy float32, written from the reference text ----
FIT = [(2.77242, 1.87083, -0.301375), (1.3841, 0.180661, -0.0212517), (0.307389, 0.170755, -0.243719), (93.4129, 43.8602, -0.556471),
       (0.597371, 0.0823258, 0.236463), (106.897, 36.4931, 0.597537), (31.4138, 7.54534, -0.881755), (116.222, 38.3149, 0.582615),
       (9.90148, 0.987682, 0.304049), (0.358194, 0.0710801, 0.0264384), (54.1034, 16.0475, -0.571797)]


def _total_key(x):
    b = int(np.array([x], f32).view(np.int32)[0])
    return b ^ (((b >> 31) & 0xFFFFFFFF) >> 1)


def _np_default(hyb, gr, qlen, graph_w=f32(0.3), hybrid_w=f32(0.6) + f32(0.1)):
    comp = {}
    for u, b, v in hyb:
        comp[u] = (f32(b), f32(v))
    clamp01 = lambda x: f32(min(max(x, f32(0)), f32(1)))
    max_act = f32(0)
    for _, a in gr:
        max_act = max(max_act, f32(a))
    g_max = max_act
    max_act = max(max_act, f32(1e-6))
    max_vec = max([f32(0)] + [v for _, v in comp.values()]); max_vec = max(max_vec, f32(1e-6))
    max_bm = max([f32(0)] + [b for b, _ in comp.values()]); max_bm = max(max_bm, f32(1e-6))
    srt = lambda xs: sorted(xs, key=lambda t: (-_total_key(t[1]), t[0]))
    by_vec = srt([(u, v) for u, (b, v) in comp.items() if v > 0])
    by_bm = srt([(u, b) for u, (b, v) in comp.items() if b > 0])

    def peak(xs):
        if not xs:
            return f32(1)
        s = f32(0)
        for _, x in xs:
            s = f32(s + x)
        mean = f32(s / f32(len(xs)))
        return f32(xs[0][1] / mean) if mean > f32(1e-6) else f32(1)
    if not by_vec or not by_bm:
        agreement = f32(0)
    else:
        k10 = max(min(10, len(by_vec), len(by_bm)), 1)
        top_v = {u for u, _ in by_vec[:k10]}
        agreement = f32(f32(sum(1 for u, _ in by_bm[:k10] if u in top_v)) / f32(k10))
    feats = [peak(by_bm), peak(by_vec), agreement, max_bm, max_vec, f32(len(by_bm)), f32(len(by_vec)), f32(len(comp)), f32(len(gr)),
             g_max, f32(qlen)]
    s = f32(-0.985886)
    for (mu, sd, w), x in zip(FIT, feats):
        s = f32(s + f32(f32(f32(w) * f32(x - f32(mu))) / f32(sd)))
    s = min(max(s, f32(-30)), f32(30))
    t = f32(f32(1) / f32(f32(1) + np.exp(f32(-s), dtype=f32)))
    trust = max(f32(f32(1) + f32(f32(2.0 - 1.0) * f32(f32(f32(2) * t) - f32(1)))), f32(0.2))
    fused = {}
    for u, a in gr:
        a = f32(a)
        sc = f32(graph_w * clamp01(f32(a / max_act)))
        fused[u] = f32(fused.get(u, f32(0)) + sc)
        fused[u] = f32(fused[u] * f32(f32(1) + f32(f32(graph_w * f32(0.3)) * clamp01(a))))
    for u, _, _ in hyb:
        b, v = comp[u]
        vn = f32(clamp01(f32(v / max_vec)) * trust)
        bn = clamp01(f32(b / max_bm))
        hi, lo = (vn, bn) if vn >= bn else (bn, vn)
        fused[u] = f32(fused.get(u, f32(0)) + f32(hybrid_w * f32(hi + f32(f32(0.3) * lo))))
    out = sorted(fused.items(), key=lambda t: (-_total_key(t[1]), t[0]))
    return out, trust


def test_default_mode_equals_numpy_restatement(M, oracle):
    rng = np.random.default_rng(5)
    lf = M.LegFusion()
    worst = 0.0
    for _ in range(60):
        hyb, gr, qlen = _case(rng)
        got, trust = lf.fuse(hyb, gr, qlen)
        exp, etrust = _np_default(hyb, gr, qlen)
        # the only libm call is expf in the gate: numpy's and glibc's may differ in the last place, everything else is exact,
        # so the trust (and with it the scores) is compared at 1e-6 relative
        assert abs(trust - float(etrust)) <= 1e-6 * max(1.0, abs(trust))
        gd, ed = dict(got), dict(exp)
        assert set(gd) == set(ed)
        for u in gd:
            worst = max(worst, abs(gd[u] - float(ed[u])))
            assert abs(gd[u] - float(ed[u])) <= 2e-6 * max(1.0, abs(gd[u]))
    assert worst < 1e-5


def test_behaviour_the_reference_documents(M):
    a, b, c, d = (bytes([i]) * 16 for i in (1, 2, 3, 4))
    # calibrated magnitude: the graph's best hit enters at graph_w (times its activation bonus), not at graph_w / (k + rank)
    got, _ = M.LegFusion(flat_adaptive=0).fuse([], [(a, 0.8), (b, 0.4)], 10)
    gd = dict(got)
    assert gd[a] == pytest.approx(0.3 * (1 + 0.3 * 0.3 * 0.8), rel=1e-6) and gd[b] == pytest.approx(0.3 * 0.5 * (1 + 0.3 * 0.3 * 0.4), rel=1e-6)
    # flat max-fusion: a vector-only candidate and a BM25-only candidate at their leg's maximum score the same; one that is
    # strong in both gets the consensus bonus on top
    got, trust = M.LegFusion(flat_adaptive=0).fuse([(a, 10.0, 0.0), (b, 0.0, 0.7), (c, 10.0, 0.7), (d, 5.0, 0.0)], [], 10)
    gd = dict(got)
    assert trust == 1.0 and gd[a] == gd[b] and gd[c] == pytest.approx(gd[a] * 1.3, rel=1e-6) and gd[d] == pytest.approx(gd[a] / 2, rel=1e-6)
    assert [u for u, _ in got] == [c, a, b, d]                                         # ties broken by uuid ascending
    # legacy RRF (SHODH_FUSION_RRF): weight / (k + rank), rank 1-based
    got, _ = M.LegFusion(fusion_rrf=1).fuse([(a, 1.0, 1.0), (b, 1.0, 1.0)], [], 0)
    assert dict(got)[a] == pytest.approx(0.7 / 31, rel=1e-6) and dict(got)[b] == pytest.approx(0.7 / 32, rel=1e-6)
    # symmetric gate stays inside [0.2, trust_max]
    rng = np.random.default_rng(2)
    for _ in range(30):
        hyb, gr, qlen = _case(rng)
        _, t = M.LegFusion().fuse(hyb, gr, qlen)
        assert 0.2 <= t <= 2.0
    # SHODH_* parsing
    lf = M.LegFusion.from_env({"SHODH_FUSION_RRF": "true", "SHODH_FLAT_ADAPTIVE": "0", "SHODH_LEG": "bm25", "SHODH_ADAPT_FEATURE": "Agreement",
                               "SHODH_GRAPH_FUSION_WEIGHT": "0"})
    assert (lf.cfg.fusion_rrf, lf.cfg.flat_adaptive, lf.cfg.isolate_leg, lf.cfg.adapt_feature, lf.cfg.graph_w) == (1, 0, 2, 1, 0.0)


def test_density_weights(M, oracle):
    # graph_retrieval.rs:2447-2472
    s, g, l = M.calculate_density_weights(0.3)
    assert abs(g - 0.5) < 1e-3 and abs(l - 0.15) < 1e-3 and abs(s + g + l - 1) < 1e-3
    s, g, l = M.calculate_density_weights(2.5)
    assert abs(g - 0.1) < 1e-3 and abs(l - 0.15) < 1e-3 and abs(s + g + l - 1) < 1e-3
    s, g, l = M.calculate_density_weights(1.25)
    assert 0.1 < g < 0.5 and abs(l - 0.15) < 1e-3 and abs(s + g + l - 1) < 1e-3
    for d in np.linspace(-1, 3, 41):
        assert np.array(M.calculate_density_weights(float(d)), f32).tobytes() == oracle.density_weights(float(d)).tobytes()
    for dens, ov, fl in itertools.product([None, 0.2, 1.0, 3.0], [None, 0.0, 0.4, 7.0], [None, 0.2, 0.6, 0.99]):
        lf = M.LegFusion(graph_density=dens, graph_weight_override=ov, graph_w_floor=fl)
        eg, eh = oracle.leg_fusion_weights(dens, ov, fl)
        assert f32(lf.cfg.graph_w).tobytes() == eg.tobytes() and f32(lf.cfg.hybrid_w).tobytes() == eh.tobytes(), (dens, ov, fl)
    assert M.LegFusion().cfg.graph_w == f32(0.3) and M.LegFusion().cfg.hybrid_w == f32(0.6) + f32(0.1)
