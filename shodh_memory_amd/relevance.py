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
