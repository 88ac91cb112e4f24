"""CPU checks of the INT8 restatement (oracle/int8_ref.py): DynamicQuantizeLinear against the three worked examples of the ONNX
operator specification (onnx/backend/test/case/node/dynamicquantizelinear.py; literals restated), the weight quantiser's
properties, MatMulInteger against a plain integer loop, and the committed fixture against a fresh run of its generator.
Parity with model_quint8_avx2.onnx itself is unpinned (no ONNX Runtime, no checkpoint offline)."""
import os

import numpy as np

from oracle import int8_ref as R

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
f32 = np.float32


def test_dynamic_quantize_linear_onnx_spec_examples():
    q, s, zp = R.dynamic_quantize(np.array([0, 2, -3, -2.5, 1.34, 0.5], f32))
    assert zp == 153 and abs(s - 0.0196078438) < 1e-9 and q.tolist() == [153, 255, 0, 26, 221, 179]
    q, s, zp = R.dynamic_quantize(np.array([-1.0, -2.1, -1.3, -2.5, -3.34, -4.0], f32))          # max adjusted to 0
    assert zp == 255 and abs(s - 0.0156862754) < 1e-9 and q.tolist() == [191, 121, 172, 96, 42, 0]
    q, s, zp = R.dynamic_quantize(np.array([[1, 2.1, 1.3, 2.5], [3.34, 4.0, 1.5, 2.6], [3.9, 4.0, 3.0, 2.345]], f32))   # min adjusted to 0
    assert zp == 0 and abs(s - 0.0156862754) < 1e-9 and q.tolist() == [[64, 134, 83, 159], [213, 255, 96, 166], [249, 255, 191, 149]]
    q, s, zp = R.dynamic_quantize(np.zeros(7, f32))                                                # degenerate range
    assert s == 1.0 and zp == 0 and not q.any()


def test_weight_quantiser_is_symmetric_8bit():
    rng = np.random.default_rng(0)
    w = rng.standard_normal((64, 48)).astype(f32)
    q, s = R.quantize_weight(w)
    assert q.dtype == np.int8 and abs(float(s) - 2 * np.abs(w).max() / 255) < 1e-9
    assert np.abs(q.astype(f32) * s - w).max() <= s / 2 + 1e-7                                      # nearest level
    assert int(np.abs(q.astype(np.int32)).max()) in (127, 128)
    assert R.quantize_weight(np.zeros((3, 3), f32))[1] == 1.0


def test_matmul_integer_is_exact():
    rng = np.random.default_rng(1)
    a = rng.integers(0, 256, (7, 1536)).astype(np.uint8)
    w = rng.integers(-128, 128, (5, 1536)).astype(np.int8)
    got = R.matmul_integer(a, 131, w)
    exp = np.array([[sum((int(a[i, k]) - 131) * int(w[j, k]) for k in range(1536)) for j in range(5)] for i in range(7)])
    assert np.array_equal(got, exp)


def test_fixture_is_what_the_generator_produces():
    from shodh_memory_amd import embedder as E
    g = np.load(os.path.join(ROOT, "tests", "golden", "encoder_int8_golden.npz"))
    sd = E.blob_to_state_dict(E.synthetic_weights(1234))
    tr = {}
    emb = R.encode(sd, g["ids"], g["mask"], trace=tr)
    assert np.array_equal(tr["acc_q"][:8, :8], g["acc_q_corner"]) and int(tr["acc_q"].astype(np.int64).sum()) == int(g["acc_q_sum"])
    assert tr["a_zp"] == int(g["a_zp"]) and f32(tr["a_scale"]) == g["a_scale"] and f32(tr["w_scale"]) == g["w_scale"]
    assert np.abs(emb - g["emb"]).max() < 1e-5 and not emb[3].any()                               # (BLAS summation order may differ between hosts)
    assert np.allclose(np.linalg.norm(emb[:3], axis=1), 1, atol=1e-5)


def test_per_text_scope_is_the_single_text_call_repeated():
    """per_text=True restates N x encode() (one session.run on [1, max_len] each, minilm.rs:883-982): the result of a row never
    depends on its batch mates, and differs from the batch tensor's (minilm.rs:996-1115), whose ranges span every row."""
    from shodh_memory_amd import embedder as E
    cfg = E.embed_cfg(layers=1, vocab=500, max_len=32)
    sd = E.blob_to_state_dict(E.synthetic_weights(5, cfg), cfg)
    rng = np.random.default_rng(2)
    ids = np.zeros((4, 32), np.int64); mask = np.zeros((4, 32), np.int64)
    for i, ln in enumerate((32, 3, 0, 17)):
        ids[i, :ln] = rng.integers(1, 500, ln); mask[i, :ln] = 1
    each = R.encode(sd, ids, mask, layers=1, per_text=True)
    for i in range(4):
        assert np.array_equal(each[i:i + 1], R.encode(sd, ids[i:i + 1], mask[i:i + 1], layers=1))
    assert np.array_equal(R.encode(sd, ids[[3, 0]], mask[[3, 0]], layers=1, per_text=True), each[[3, 0]])
    batch = R.encode(sd, ids, mask, layers=1)
    assert not np.array_equal(batch, each) and not each[2].any() and not batch[2].any()
    keep = [0, 1, 3]
    assert ((batch[keep] * each[keep]).sum(1) > 0.98).all()
