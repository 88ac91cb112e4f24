"""A second, HARSH synthetic weight set for the encoder parity tests (test infrastructure).

The library's own generator (shodh_embedder_synthetic_weights) draws every matrix from N(0, 0.02) with LayerNorm gamma 1 / beta 0:
well-conditioned, unlike a trained checkpoint. Real BERT-family models have a few hidden dimensions that run 20-50x hotter than the
rest, LayerNorm gains spread over two decades and a heavy-tailed word table -- exactly what stresses bf16 activations and 8-bit
per-tensor ranges. The real all-MiniLM-L6-v2 weights cannot be fetched offline, so this set imitates those properties, from a seed:
numpy's PCG64 stream is stable across versions, so the fixture script (tests/golden/make_encoder_golden.py, on a CPU with
transformers) and the GPU box regenerate identical bytes; nothing but the seed is stored."""
import numpy as np

f32 = np.float32
OUTLIER_DIMS = (7, 111, 200, 333)


def harsh_state_dict(E, seed=4242, cfg=None):
    cfg = cfg or E.embed_cfg()
    rng = np.random.default_rng(seed)
    H, I = cfg.hidden, cfg.intermediate
    sd = {k: np.zeros_like(v) for k, v in E.blob_to_state_dict(np.zeros(E.param_count(cfg), f32), cfg).items()}
    out = list(OUTLIER_DIMS)

    def normal(shape, std):
        return (rng.standard_normal(shape) * std).astype(f32)

    def gains(n):                                            # log-uniform over [0.1, 10]
        return np.exp(rng.uniform(np.log(0.1), np.log(10.0), n)).astype(f32)
    word = (rng.standard_t(3, (cfg.vocab, H)) * 0.15).astype(f32)       # heavy tails (df = 3): a few entries at 10-30 sigma
    word = np.clip(word, -8, 8)
    word[:, out] *= f32(25.0)                                # outlier channels
    sd["embeddings.word_embeddings.weight"] = word
    sd["embeddings.position_embeddings.weight"] = normal((cfg.max_pos, H), 0.05)
    sd["embeddings.token_type_embeddings.weight"] = normal((cfg.type_vocab, H), 0.05)
    sd["embeddings.LayerNorm.weight"] = gains(H)
    sd["embeddings.LayerNorm.bias"] = normal(H, 0.3)
    for l in range(cfg.layers):
        p = "encoder.layer.%d." % l
        for nm in ("query", "key", "value"):
            sd[p + "attention.self.%s.weight" % nm] = normal((H, H), 0.06)
            sd[p + "attention.self.%s.bias" % nm] = normal(H, 0.2)
        wo = normal((H, H), 0.04)
        wo[out, :] *= f32(6.0)                               # the outlier dimensions are fed by hot output rows, layer after layer
        sd[p + "attention.output.dense.weight"] = wo
        sd[p + "attention.output.dense.bias"] = normal(H, 0.1)
        g1 = gains(H); g1[out] = f32(10.0)
        sd[p + "attention.output.LayerNorm.weight"] = g1
        b1 = normal(H, 0.3); b1[out] += f32(4.0)
        sd[p + "attention.output.LayerNorm.bias"] = b1
        sd[p + "intermediate.dense.weight"] = normal((I, H), 0.04)
        sd[p + "intermediate.dense.bias"] = normal(I, 0.3)
        wd = normal((H, I), 0.03)
        wd[out, :] *= f32(6.0)
        sd[p + "output.dense.weight"] = wd
        sd[p + "output.dense.bias"] = normal(H, 0.1)
        g2 = gains(H); g2[out] = f32(8.0)
        sd[p + "output.LayerNorm.weight"] = g2
        sd[p + "output.LayerNorm.bias"] = normal(H, 0.3)
    return sd


def harsh_blob(E, seed=4242, cfg=None):
    return E.state_dict_to_blob(harsh_state_dict(E, seed, cfg), cfg)
