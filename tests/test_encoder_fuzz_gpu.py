"""Randomised encoder check: random batch sizes (1 .. 3000 texts), lengths (0 .. 128 incl. empty and full), the bf16 path (fused
feed-forward kernel, MFMA attention) against the fp32 path of the same library (itself pinned to transformers.BertModel by
test_encoder_gpu.py): cosine >= 0.999 per text, zero vectors for empty texts, bit-identical repeats. SHODH_FUZZ_ROUNDS (default 6)."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def S():
    import shodh_memory_amd as s
    return s


def test_random_encoder_batches(S):
    from shodh_memory_amd import _lib as L
    rounds = int(os.environ.get("SHODH_FUZZ_ROUNDS", "6"))
    rng = np.random.default_rng(int(os.environ.get("SHODH_FUZZ_SEED", "31337")))
    bf = S.MiniLMEmbedder(synthetic_seed=99, dtype=L.DTYPE_BF16)
    fp = S.MiniLMEmbedder(synthetic_seed=99, dtype=L.DTYPE_FP32)
    ML = 256
    for rnd in range(rounds):
        b = int(rng.choice([1, 2, 3, 31, 64, 257, 1000, 3000]))
        mode = int(rng.integers(0, 4))
        if mode == 0:
            lens = rng.integers(0, 129, b)
        elif mode == 1:
            lens = np.full(b, int(rng.choice([1, 2, 32, 33, 127, 128])))
        elif mode == 2:
            lens = rng.choice([0, 1, 128], b)
        else:
            lens = rng.integers(1, 20, b)
        print("encoder fuzz round %d: batch %d mode %d tokens %d" % (rnd, b, mode, int(lens.sum())), flush=True)
        ids = np.zeros((b, ML), np.int32); mask = np.zeros((b, ML), np.uint8)
        for i, n in enumerate(lens):
            ids[i, :n] = rng.integers(1000, 30522, n); mask[i, :n] = 1
        a = bf.encode_ids(ids, mask)
        a2 = bf.encode_ids(ids, mask)
        r = fp.encode_ids(ids, mask)
        assert a.tobytes() == a2.tobytes()                                   # deterministic: no race shows as run-to-run noise
        assert np.isfinite(a).all()
        empty = lens == 0
        assert (a[empty] == 0).all() and (r[empty] == 0).all()
        if (~empty).any():
            cos = (a[~empty] * r[~empty]).sum(1)
            assert cos.min() >= 0.999, (rnd, float(cos.min()), int(np.argmin(cos)))
            assert np.allclose(np.linalg.norm(a[~empty], axis=1), 1.0, atol=2e-3)


def test_random_int8_batches_fused_layer_vs_round2_kernels(S, monkeypatch):
    """The fused INT8 layer (per-sequence q|k|v + attention with in-kernel quantisation, streaming GEMMs with LayerNorm / GELU epilogues, K-tiled FFN
    down) against the round-2 kernels (SHODH_INT8_STAGES=0) on random batches: batch sizes around the tile sizes, lengths incl. empty / one token /
    the full window (exactly 4 key blocks), padded and unpadded tensors. Same function, so cosine >= 0.9999 per text (the difference is f32
    reassociation inside a stage and the f16-split attention products), zero vectors for empty texts, bit-identical repeats."""
    from shodh_memory_amd import _lib as L
    rounds = int(os.environ.get("SHODH_FUZZ_ROUNDS", "6"))
    rng = np.random.default_rng(int(os.environ.get("SHODH_FUZZ_SEED", "424242")))
    ML = 256
    for rnd in range(rounds):
        padded = bool(rng.integers(0, 2))
        monkeypatch.delenv("SHODH_INT8_STAGES", raising=False)
        fused = S.MiniLMEmbedder(synthetic_seed=7 + rnd, dtype=L.DTYPE_INT8, compute_padded=padded)
        monkeypatch.setenv("SHODH_INT8_STAGES", "0")
        old = S.MiniLMEmbedder(synthetic_seed=7 + rnd, dtype=L.DTYPE_INT8, compute_padded=padded)
        b = int(rng.choice([1, 2, 3, 7, 31, 64, 65, 200]))
        mode = int(rng.integers(0, 4))
        if mode == 0:
            lens = rng.integers(0, 129, b)
        elif mode == 1:
            lens = np.full(b, int(rng.choice([1, 2, 31, 32, 33, 96, 97, 127, 128])))
        elif mode == 2:
            lens = rng.choice([0, 1, 128], b)
        else:
            lens = rng.integers(1, 20, b)
        print("int8 fuzz round %d: batch %d mode %d padded %s tokens %d" % (rnd, b, mode, padded, int(lens.sum())), flush=True)
        ids = np.zeros((b, ML), np.int32); mask = np.zeros((b, ML), np.uint8)
        for i, n in enumerate(lens):
            ids[i, :n] = rng.integers(1000, 30522, n); mask[i, :n] = 1
        a = fused.encode_ids(ids, mask)
        a2 = fused.encode_ids(ids, mask)
        r = old.encode_ids(ids, mask)
        fused.close(); old.close()
        assert a.tobytes() == a2.tobytes() and np.isfinite(a).all()
        empty = lens == 0
        assert (a[empty] == 0).all() and (r[empty] == 0).all()
        if (~empty).any():
            cos = (a[~empty] * r[~empty]).sum(1)
            assert cos.min() >= 0.9999, (rnd, float(cos.min()), int(np.argmin(cos)))
            assert np.allclose(np.linalg.norm(a[~empty], axis=1), 1.0, atol=2e-3)


def test_random_int8_batches_with_weight_zero_points(S, tmp_path, monkeypatch):
    """The same comparison on a model loaded from a file written like the dynamic-quantisation export (uint8 weights, a zero point per output channel):
    every fused kernel then carries the  - z[n] * rowsum_a[m]  term -- the row sums travel with the tile DMA (streaming kernels), come from the wave's own
    fragments (FFN down) or from the in-kernel quantisation (q|k|v) -- on batches whose token counts are not multiples of the 64-token tiles."""
    from oracle import int8_ref as R
    from shodh_memory_amd import _lib as L
    from shodh_memory_amd import embedder as E
    from tests import onnx_writer as W
    rounds = int(os.environ.get("SHODH_FUZZ_ROUNDS", "5"))
    rng = np.random.default_rng(int(os.environ.get("SHODH_FUZZ_SEED", "77")))
    cfg = E.embed_cfg(layers=2, vocab=3000)
    sd = E.blob_to_state_dict(E.synthetic_weights(11, cfg), cfg)
    for k in sd:
        if k.endswith("dense.weight") or k.endswith("query.weight") or k.endswith("value.weight"):
            sd[k] += np.float32(0.004)
    qm = R.quantize_model(sd, cfg.layers, rule=lambda w: R.quantize_weight_ort(w, per_channel=True), word_rule=lambda w: R.quantize_weight_ort(w))
    path = str(tmp_path / "model_quantized.onnx")
    W.write_bert(path, sd, cfg.layers, qmodel=qm)
    ML = 256
    for rnd in range(rounds):
        padded = bool(rng.integers(0, 2))
        monkeypatch.delenv("SHODH_INT8_STAGES", raising=False)
        fused = S.MiniLMEmbedder(dtype=L.DTYPE_INT8, weights_path=path, layers=cfg.layers, vocab=cfg.vocab, compute_padded=padded)
        monkeypatch.setenv("SHODH_INT8_STAGES", "0")
        old = S.MiniLMEmbedder(dtype=L.DTYPE_INT8, weights_path=path, layers=cfg.layers, vocab=cfg.vocab, compute_padded=padded)
        b = int(rng.choice([1, 2, 3, 7, 33, 70, 130]))
        lens = rng.integers(0, 129, b) if rnd % 2 else rng.choice([1, 5, 31, 64, 65, 128], b)
        print("int8 zero-point fuzz round %d: batch %d padded %s tokens %d" % (rnd, b, padded, int(lens.sum())), flush=True)
        ids = np.zeros((b, ML), np.int32); mask = np.zeros((b, ML), np.uint8)
        for i, n in enumerate(lens):
            ids[i, :n] = rng.integers(5, cfg.vocab, n); mask[i, :n] = 1
        a = fused.encode_ids(ids, mask)
        a2 = fused.encode_ids(ids, mask)
        r = old.encode_ids(ids, mask)
        fused.close(); old.close()
        assert a.tobytes() == a2.tobytes() and np.isfinite(a).all()
        empty = lens == 0
        assert (a[empty] == 0).all() and (r[empty] == 0).all()
        if (~empty).any():
            cos = (a[~empty] * r[~empty]).sum(1)
            assert cos.min() >= 0.9999, (rnd, float(cos.min()), int(np.argmin(cos)))
