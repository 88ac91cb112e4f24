"""GPU tests of the INT8 encoder mode (SHODH_DTYPE_INT8; csrc/encoder_int8.h) through the C ABI.
Gates: the dense layer's quantisation parameters and int32 accumulators are EXACT against the numpy
restatement of DynamicQuantizeLinear / MatMulInteger (oracle/int8_ref.py), its float output within 1e-6 relative; the whole
encoder agrees with the restatement to cosine >= 0.9999 and with the fp32 encoder to cosine >= 0.98 (the reference's own figure
for quantisation-level differences is 0.9859, minilm.rs:591). Parity with model_quint8_avx2.onnx is UNPINNED."""
import ctypes as C
import os

import numpy as np
import pytest

from oracle import int8_ref as R

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
f32 = np.float32


@pytest.fixture(scope="module")
def S():
    from shodh_memory_amd import build
    build.build()
    import shodh_memory_amd as s
    return s


def int8_dense(x, w, bias):
    from shodh_memory_amd import _lib as L
    M, K = x.shape
    N = w.shape[0]
    y = np.zeros((M, N), f32); acc = np.zeros((M, N), np.int32)
    sa, zp, sw = C.c_float(), C.c_int32(), C.c_float()
    L.check(L.lib().shodh_int8_dense(0, x.ctypes.data, w.ctypes.data, bias.ctypes.data if bias is not None else None, M, N, K,
                                     y.ctypes.data, acc.ctypes.data, C.byref(sa), C.byref(zp), C.byref(sw)))
    return y, acc, f32(sa.value), int(zp.value), f32(sw.value)


def cos(a, b):
    return (a * b).sum(1) / np.maximum(np.linalg.norm(a, axis=1) * np.linalg.norm(b, axis=1), 1e-30)


@pytest.mark.parametrize("M,N,K,kind", [(300, 256, 384, "normal"), (1, 128, 128, "normal"), (513, 384, 1536, "gelu-like"), (200, 128, 384, "negative"),
                                        (64, 128, 256, "zeros"), (129, 1536, 384, "outlier")])
def test_int8_dense_exact_accumulators(S, M, N, K, kind):
    rng = np.random.default_rng(M + N + K)
    x = rng.standard_normal((M, K)).astype(f32)
    if kind == "gelu-like":
        x = np.maximum(x, f32(-0.17)) * f32(3)                       # one-sided range: the zero point sits near 0
    elif kind == "negative":
        x = -np.abs(x) - f32(0.5)                                    # max adjusted up to 0: zero point 255
    elif kind == "zeros":
        x[:] = 0
    elif kind == "outlier":
        x[7, 5] = f32(80.0)                                          # one outlier stretches the per-tensor range
    w = (rng.standard_normal((N, K)) * 0.05).astype(f32)
    b = rng.standard_normal(N).astype(f32)
    y, acc, sa, zp, sw = int8_dense(x, w, b)
    wq, ws = R.quantize_weight(w)
    e_y, e_acc, e_sa, e_zp = R.dense_int8(x, wq, ws, b)
    assert sw == ws and sa == e_sa and zp == e_zp, (sw, ws, sa, e_sa, zp, e_zp)
    assert np.array_equal(acc, e_acc), np.argwhere(acc != e_acc)[:5]                  # int32 accumulators: bit for bit
    assert np.abs(y - e_y).max() <= 1e-6 * max(1.0, float(np.abs(e_y).max()))


def test_int8_encoder_matches_restatement_and_fixture(S):
    from shodh_memory_amd import _lib as L
    from shodh_memory_amd import embedder as E
    g = np.load(os.path.join(ROOT, "tests", "golden", "encoder_int8_golden.npz"))
    e8 = S.MiniLMEmbedder(synthetic_seed=1234, dtype=L.DTYPE_INT8)                    # compute_padded = 1: the reference's padded tensor
    emb = e8.encode_ids(g["ids"], g["mask"])
    assert not emb[3].any()                                                           # empty text -> zero vector
    c = cos(emb[:3], g["emb"][:3])
    assert c.min() >= 0.9999, c
    assert np.abs(emb[:3] - g["emb"][:3]).max() < 2e-3
    assert np.allclose(np.linalg.norm(emb[:3], axis=1), 1, atol=1e-4)
    # against the fp32 encoder: quantisation-level agreement
    e32 = S.MiniLMEmbedder(synthetic_seed=1234, dtype=L.DTYPE_FP32)
    ref = e32.encode_ids(g["ids"], g["mask"])
    c32 = cos(emb[:3], ref[:3])
    print("INT8 vs fp32 cosine:", c32)
    assert c32.min() >= 0.98, c32
    # the batch is ONE tensor: DynamicQuantizeLinear ranges span all its rows, so a row's embedding depends (slightly) on its
    # batch mates -- as in the reference, where encode_batch runs one session.run per batch (minilm.rs:996-1115)
    alone = e8.encode_ids(g["ids"][1:2], g["mask"][1:2])
    sd = E.blob_to_state_dict(E.synthetic_weights(1234))
    exp_alone = R.encode(sd, g["ids"][1:2], g["mask"][1:2])
    assert cos(alone, exp_alone).min() >= 0.9999 and cos(alone, emb[1:2]).min() >= 0.98
    # unpadded variant (compute_padded = 0): a different, cheaper function -- ranges over the real tokens only
    e8u = S.MiniLMEmbedder(synthetic_seed=1234, dtype=L.DTYPE_INT8, compute_padded=False)
    un = e8u.encode_ids(g["ids"], g["mask"])
    assert cos(un[:3], ref[:3]).min() >= 0.98 and not un[3].any()


def test_int8_encoder_device_api_and_larger_batch(S):
    import torch
    from shodh_memory_amd import _lib as L
    from tests import bert_ref
    e8 = S.MiniLMEmbedder(synthetic_seed=1234, dtype=L.DTYPE_INT8)
    e32 = S.MiniLMEmbedder(synthetic_seed=1234, dtype=L.DTYPE_FP32)
    ids, mask = bert_ref.synth_batch(48, 256, seed=3)
    d_ids = ids.to(torch.int32).cuda().contiguous(); d_mask = mask.to(torch.uint8).cuda().contiguous()
    out = e8.encode_ids_device(d_ids, d_mask)
    torch.cuda.synchronize()
    ref = e32.encode_ids(ids.numpy().astype(np.int32), mask.numpy().astype(np.uint8))
    c = cos(out.cpu().numpy(), ref)
    assert c.min() >= 0.98, c.min()
    with pytest.raises(L.ShodhError):
        S.MiniLMEmbedder(synthetic_seed=1234, dtype=L.DTYPE_BF16, compute_padded=True)      # padding only matters for INT8
