"""LearnedWeights (src/relevance.rs:343-606) over the C ABI: same method names and semantics."""
import ctypes as C

import numpy as np

from . import _lib as L


class LearnedWeights:
    def __init__(self, semantic=None, entity=None, tag=None, importance=None, momentum=None,
                 access_count=None, graph_strength=None):
        self._w = L.Weights()
        L.lib().shodh_weights_default(C.byref(self._w))      # impl Default (relevance.rs:383-397)
        for name, v in dict(semantic=semantic, entity=entity, tag=tag, importance=importance, momentum=momentum,
                            access_count=access_count, graph_strength=graph_strength).items():
            if v is not None:
                setattr(self._w, name, v)

    @classmethod
    def default(cls):
        return cls()

    def __getattr__(self, name):
        if name in ("semantic", "entity", "tag", "importance", "momentum", "access_count", "graph_strength", "update_count"):
            return getattr(self._w, name)
        raise AttributeError(name)

    def as_tuple(self):
        w = self._w
        return (w.semantic, w.entity, w.tag, w.importance, w.momentum, w.access_count, w.graph_strength)

    def normalize(self):
        L.lib().shodh_weights_normalize(C.byref(self._w))

    def apply_feedback(self, semantic_contributed, entity_contributed, tag_contributed, helpful):
        L.lib().shodh_weights_apply_feedback(C.byref(self._w), int(semantic_contributed), int(entity_contributed),
                                             int(tag_contributed), int(helpful))

    def fuse_scores(self, semantic_score, entity_score, tag_score, importance_score):
        return float(L.lib().shodh_fuse_scores(C.byref(self._w), semantic_score, entity_score, tag_score, importance_score))

    def fuse_scores_with_momentum(self, semantic_score, entity_score, tag_score, importance_score, momentum_ema):
        return float(L.lib().shodh_fuse_scores_with_momentum(C.byref(self._w), semantic_score, entity_score, tag_score,
                                                             importance_score, momentum_ema))

    def fuse_scores_full(self, semantic_score, entity_score, tag_score, importance_score, momentum_ema, access_count, graph_strength):
        return float(L.lib().shodh_fuse_scores_full(C.byref(self._w), semantic_score, entity_score, tag_score, importance_score,
                                                    momentum_ema, int(access_count), graph_strength))

    def fuse_scores_full_batch(self, sem, ent, tag, imp, mom, acc, gs, device=0):
        arrs = [np.ascontiguousarray(a, np.float32) for a in (sem, ent, tag, imp, mom)]
        acc = np.ascontiguousarray(acc, np.uint32)
        gs = np.ascontiguousarray(gs, np.float32)
        n = arrs[0].size
        out = np.zeros(n, np.float32)
        L.check(L.lib().shodh_fuse_scores_full_batch(device, C.byref(self._w), n, *[a.ctypes.data for a in arrs],
                                                     acc.ctypes.data, gs.ctypes.data, out.ctypes.data))
        return out


def calibrate_score(score):
    return float(L.lib().shodh_calibrate_score(score))


# ---- recall Layer 4: hybrid (vector + BM25) leg + graph leg fusion (src/memory/mod.rs:3878-4468) ----
def calculate_density_weights(graph_density):
    """graph_retrieval.rs:81-101 -> (semantic_weight, graph_weight, linguistic_weight)"""
    out = np.zeros(3, np.float32)
    L.lib().shodh_density_weights(graph_density, out.ctypes.data)
    return float(out[0]), float(out[1]), float(out[2])


class LegFusion:
    """The fusion step after `search_ids` on the recall path. The reference reads its switches from SHODH_* environment
    variables inside `MemorySystem::recall`; `from_env()` does the same, the constructor takes them as keywords
    (field names of `shodh_leg_fusion_cfg`)."""
    _FEATURES = {"fitted": 0, "agreement": 1}          # anything else is the peakedness feature (mod.rs:4206)
    _LEGS = {"vector": 1, "bm25": 2, "graph": 3}

    def __init__(self, graph_density=None, graph_weight_override=None, graph_w_floor=None, **fields):
        self.cfg = L.LegFusionCfg()
        L.lib().shodh_leg_fusion_cfg_default(C.byref(self.cfg))
        g, h = C.c_float(), C.c_float()
        nan = float("nan")
        L.lib().shodh_leg_fusion_weights(int(graph_density is not None), 0.0 if graph_density is None else graph_density,
                                         nan if graph_weight_override is None else graph_weight_override,
                                         nan if graph_w_floor is None else graph_w_floor, C.byref(g), C.byref(h))
        self.cfg.graph_w, self.cfg.hybrid_w = g.value, h.value
        names = {f[0] for f in L.LegFusionCfg._fields_}
        for k, v in fields.items():
            if k not in names:
                raise TypeError(f"unknown fusion field {k!r}")
            setattr(self.cfg, k, v)

    @classmethod
    def from_env(cls, env, graph_density=None):
        """env: a mapping like os.environ. Parsing as the reference: flags are "1"/"true" (on-by-default ones: not "0"/"false")."""
        def on(key):
            v = env.get(key)
            return v is not None and (v == "1" or v.lower() == "true")

        def not_off(key):
            v = env.get(key)
            return not (v is not None and (v == "0" or v.lower() == "false"))

        def num(key, default):
            try:
                return float(env[key])
            except (KeyError, ValueError):
                return default
        fields = dict(fusion_v2=on("SHODH_FUSION_V2"), fusion_flat=on("SHODH_FUSION_FLAT"), fusion_sum=on("SHODH_FUSION_SUM"),
                      fusion_rrf=on("SHODH_FUSION_RRF"), isolate_leg=cls._LEGS.get(env.get("SHODH_LEG", ""), 0),
                      flat_adaptive=not_off("SHODH_FLAT_ADAPTIVE"), adapt_symmetric=not_off("SHODH_ADAPT_SYMMETRIC"),
                      adapt_feature=cls._FEATURES.get(env.get("SHODH_ADAPT_FEATURE", "fitted").lower(), 2),
                      flat_consensus=num("SHODH_FLAT_CONSENSUS", 0.3), adapt_trust_max=num("SHODH_ADAPT_TRUST_MAX", 2.0),
                      fw_graph=num("SHODH_FW_GRAPH", 0.3), fw_vec=num("SHODH_FW_VEC", 0.6), fw_bm25=num("SHODH_FW_BM25", 0.4),
                      agree_k=num("SHODH_ADAPT_AGREE_K", 10.0), agree_lo=num("SHODH_ADAPT_AGREE_LO", 0.1),
                      agree_hi=num("SHODH_ADAPT_AGREE_HI", 0.5), peak_lo=num("SHODH_ADAPT_PEAK_LO", 2.0), peak_hi=num("SHODH_ADAPT_PEAK_HI", 6.0))
        return cls(graph_density=graph_density,
                   graph_weight_override=num("SHODH_GRAPH_FUSION_WEIGHT", None), graph_w_floor=num("SHODH_GRAPH_W_FLOOR", None), **fields)

    def fuse(self, hybrid, graph, query_len):
        """hybrid: [(uuid bytes16, bm25_score, vector_score)] in hybrid rank order; graph: [(uuid bytes16, activation)] in rank
        order. Returns ([(uuid, fused score)] sorted score desc / uuid asc, effective_vec_trust)."""
        hu = np.frombuffer(b"".join(h[0] for h in hybrid), np.uint8).copy() if hybrid else np.zeros(16, np.uint8)
        hb = np.array([h[1] for h in hybrid] or [0], np.float32)
        hv = np.array([h[2] for h in hybrid] or [0], np.float32)
        gu = np.frombuffer(b"".join(g[0] for g in graph), np.uint8).copy() if graph else np.zeros(16, np.uint8)
        ga = np.array([g[1] for g in graph] or [0], np.float32)
        cap = len(hybrid) + len(graph) + 1
        ou = np.zeros((cap, 16), np.uint8)
        os_ = np.zeros(cap, np.float32)
        trust = C.c_float()
        m = L.lib().shodh_fuse_legs(C.byref(self.cfg), hu.ctypes.data, hb.ctypes.data, hv.ctypes.data, len(hybrid), gu.ctypes.data,
                                    ga.ctypes.data, len(graph), int(query_len), ou.ctypes.data, os_.ctypes.data, cap, C.byref(trust))
        return [(bytes(ou[i]), float(os_[i])) for i in range(m)], float(trust.value)


# ---- ranking tail of RelevanceEngine::surface_relevant_inner (src/relevance.rs:801-918) ----
REASONS = ("Combined", "EntityMatch", "SemanticSimilarity", "RecentImportant")      # RelevanceReason


def calculate_tag_score(context, tags):
    """relevance.rs:680-705"""
    arr = (C.c_char_p * max(len(tags), 1))(*[t.encode("utf-8") for t in tags])
    return float(L.lib().shodh_calculate_tag_score(context.encode("utf-8"), arr, len(tags)))


def apply_recency_boost(base_score, age_hours, boost_hours, multiplier):
    """relevance.rs:1524-1547 with age_hours = (now - created_at).num_hours()"""
    return float(L.lib().shodh_apply_recency_boost(base_score, int(age_hours), int(boost_hours), multiplier))


def rank_surfaced(weights, candidates, **config):
    """Phase 3 of surface_relevant_inner. candidates: dicts with semantic, entity, tag, importance, momentum, access_count,
    graph_strength, age_hours, created_at_ns, uuid (bytes16). config: fields of RelevanceConfig (min_importance,
    recency_boost_hours, recency_boost_multiplier, graph_boost_multiplier, max_results). -> [(index, score, reason name)]"""
    cfg = L.RelevanceCfg()
    L.lib().shodh_relevance_cfg_default(C.byref(cfg))
    for k, v in config.items():
        if k not in {f[0] for f in L.RelevanceCfg._fields_}:
            raise TypeError(f"unknown RelevanceConfig field {k!r}")
        setattr(cfg, k, v)
    n = len(candidates)
    col = lambda key, dt: np.array([c[key] for c in candidates] or [0], dt)
    sem, ent, tag, imp, mom, gs = (col(k, np.float32) for k in ("semantic", "entity", "tag", "importance", "momentum", "graph_strength"))
    acc, age, cre = col("access_count", np.uint32), col("age_hours", np.int64), col("created_at_ns", np.int64)
    uu = np.frombuffer(b"".join(c["uuid"] for c in candidates), np.uint8).copy() if n else np.zeros(16, np.uint8)
    cap = max(int(cfg.max_results), 1)
    oi, os_, orr = np.zeros(cap, np.uint32), np.zeros(cap, np.float32), np.zeros(cap, np.uint8)
    w = weights._w if isinstance(weights, LearnedWeights) else weights
    m = L.lib().shodh_rank_surfaced(C.byref(w), C.byref(cfg), n, sem.ctypes.data, ent.ctypes.data, tag.ctypes.data, imp.ctypes.data, mom.ctypes.data,
                                    acc.ctypes.data, gs.ctypes.data, age.ctypes.data, cre.ctypes.data, uu.ctypes.data, oi.ctypes.data, os_.ctypes.data,
                                    orr.ctypes.data)
    return [(int(oi[i]), float(os_[i]), REASONS[int(orr[i])]) for i in range(m)]
