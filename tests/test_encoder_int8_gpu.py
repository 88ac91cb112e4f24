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


# ---- round 3: an export's own quantised tensors (arbitrary zero points, per tensor or per channel), weight files, and the fused layer ----
def int8_dense_quantized(x, wq, w_scale, w_zp, bias):
    from shodh_memory_amd import _lib as L
    M, K = x.shape
    N = wq.shape[0]
    y = np.zeros((M, N), f32); acc = np.zeros((M, N), np.int32)
    sa, zp = C.c_float(), C.c_int32()
    sc = np.ascontiguousarray(np.atleast_1d(w_scale), f32); z = np.ascontiguousarray(np.atleast_1d(w_zp), wq.dtype)
    L.check(L.lib().shodh_int8_dense_quantized(0, x.ctypes.data, wq.ctypes.data, int(wq.dtype == np.int8), sc.ctypes.data, z.ctypes.data, sc.size,
                                               bias.ctypes.data if bias is not None else None, M, N, K, y.ctypes.data, acc.ctypes.data, C.byref(sa), C.byref(zp)))
    return y, acc, f32(sa.value), int(zp.value)


@pytest.mark.parametrize("kw", [dict(), dict(per_channel=True), dict(symmetric=True), dict(per_channel=True, reduce_range=True), dict(signed=True, per_channel=True),
                                dict(signed=True, symmetric=True)])
@pytest.mark.parametrize("M,N,K", [(300, 256, 384), (65, 384, 1536)])
def test_int8_dense_on_an_exports_tensor_exact_accumulators(S, kw, M, N, K):
    """MatMulInteger with a weight zero point of any value: sum (a - a_zp)(b - b_zp), bit for bit"""
    rng = np.random.default_rng(M + N + K + len(kw))
    x = (rng.standard_normal((M, K)) * 2 + 0.7).astype(f32)
    w = (rng.standard_normal((N, K)) * 0.05 + 0.01).astype(f32)
    w[:, 3] *= 6                                                       # an outlier input channel: the per-channel ranges differ
    b = rng.standard_normal(N).astype(f32)
    wq, ws, wz = R.quantize_weight_ort(w, **kw)
    y, acc, sa, zp = int8_dense_quantized(x, wq, ws, wz, b)
    e_y, e_acc, e_sa, e_zp = R.dense_int8(x, wq, ws, b, wz)
    assert sa == e_sa and zp == e_zp
    assert np.array_equal(acc, e_acc), np.argwhere(acc != e_acc)[:5]
    assert np.abs(y - e_y).max() <= 1e-6 * max(1.0, float(np.abs(e_y).max()))


def _export_case(E, rule, layers=2, vocab=3000, seed=77):
    from shodh_memory_amd import _lib as L
    cfg = E.embed_cfg(layers=layers, vocab=vocab)
    rng = np.random.default_rng(seed)
    blob = E.synthetic_weights(seed, cfg)
    sd = E.blob_to_state_dict(blob, cfg)
    # LayerNorm gains / biases away from 1 / 0 and a shifted dense weight: asymmetric ranges for the quantiser to see
    for k in sd:
        if k.endswith("LayerNorm.weight"):
            sd[k][:] = rng.uniform(0.5, 1.5, sd[k].shape)
        elif k.endswith("LayerNorm.bias"):
            sd[k][:] = rng.normal(0, 0.1, sd[k].shape)
        elif k.endswith("dense.weight") or k.endswith("query.weight"):
            sd[k] += f32(0.004)
    word_rule = lambda w: R.quantize_weight_ort(w)
    qm = R.quantize_model(sd, layers, rule=rule, word_rule=word_rule)
    return cfg, sd, qm


def _batch(n, seed, vocab):
    rng = np.random.default_rng(seed)
    ids = np.zeros((n, 256), np.int32); mask = np.zeros((n, 256), np.uint8)
    lens = rng.integers(1, 129, n)
    lens[0] = 128; lens[-1] = 1
    if n > 3:
        lens[2] = 0                                                    # an empty text
    for i, ln in enumerate(lens):
        ids[i, :ln] = rng.integers(1, vocab, ln); mask[i, :ln] = 1
    return ids, mask


@pytest.mark.parametrize("rule_name", ["u8_per_channel_asym", "u8_per_tensor_asym", "s8_symmetric"])
def test_int8_encoder_runs_a_quantised_onnx_export(S, tmp_path, rule_name):
    """weights_path = a dynamic-quantisation export: the file's own uint8 / int8 tensors, scales and zero points are what the device multiplies"""
    from shodh_memory_amd import _lib as L
    from shodh_memory_amd import embedder as E
    from tests import onnx_writer as W
    rules = {"u8_per_channel_asym": lambda w: R.quantize_weight_ort(w, per_channel=True), "u8_per_tensor_asym": lambda w: R.quantize_weight_ort(w),
             "s8_symmetric": lambda w: R.quantize_weight_ort(w, symmetric=True, signed=True)}
    cfg, sd, qm = _export_case(E, rules[rule_name])
    path = str(tmp_path / "model_quantized.onnx")
    W.write_bert(path, sd, cfg.layers, qmodel=qm)
    e8 = S.MiniLMEmbedder(dtype=L.DTYPE_INT8, weights_path=path, layers=cfg.layers, vocab=cfg.vocab)
    assert e8.weight_source("encoder.layer.1.output.dense.weight") == L.WEIGHT_EXPORT_Q8
    assert e8.weight_source("embeddings.word_embeddings.weight") == L.WEIGHT_EXPORT_Q8
    assert e8.weight_source("encoder.layer.0.output.dense.bias") == L.WEIGHT_F32
    ids, mask = _batch(6, 5, cfg.vocab)
    emb = e8.encode_ids(ids, mask)
    exp = R.encode(sd, ids, mask, layers=cfg.layers, qmodel=qm)
    keep = mask.sum(1) > 0
    c = cos(emb[keep], exp[keep])
    print(rule_name, "cosine vs restatement:", c)
    assert c.min() >= 0.9999 and not emb[~keep].any()
    # the same tensors handed over one by one give the same bits as the file
    e2 = S.MiniLMEmbedder(dtype=L.DTYPE_INT8, layers=cfg.layers, vocab=cfg.vocab)
    for name, a in sd.items():
        if name in qm:
            q, sc, zp = qm[name]
            e2.load_quantized(name, q, sc, zp)
        else:
            e2.load_tensor(name, a)
    with pytest.raises(L.ShodhError):
        e2.load_tensor("encoder.layer.9.nope", np.zeros(3, f32))
    e2.finish_weights()
    assert e2.encode_ids(ids, mask).tobytes() == emb.tobytes()
    # the f32 view of the same file in bf16 mode: dequantised weights, close to the INT8 result
    eb = S.MiniLMEmbedder(dtype=L.DTYPE_BF16, weights_path=path, layers=cfg.layers, vocab=cfg.vocab)
    assert eb.weight_source("encoder.layer.1.output.dense.weight") == L.WEIGHT_F32
    cb = cos(eb.encode_ids(ids, mask)[keep], emb[keep])
    assert cb.min() >= 0.97, cb


def test_safetensors_checkpoint_through_weights_path(S, tmp_path):
    from shodh_memory_amd import _lib as L
    from shodh_memory_amd import embedder as E
    from tests import onnx_writer as W
    blob = E.synthetic_weights(4321)
    path = str(tmp_path / "model.safetensors")
    W.write_safetensors(path, E.blob_to_state_dict(blob))
    ids, mask = _batch(5, 9, 30522)
    for dt in (L.DTYPE_FP32, L.DTYPE_BF16, L.DTYPE_INT8):
        a = S.MiniLMEmbedder(dtype=dt, weights_path=path).encode_ids(ids, mask)
        b = S.MiniLMEmbedder(dtype=dt, weights=blob).encode_ids(ids, mask)
        assert a.tobytes() == b.tobytes()
    assert S.MiniLMEmbedder(dtype=L.DTYPE_INT8, weights_path=path).weight_source("encoder.layer.0.intermediate.dense.weight") == L.WEIGHT_SELF_Q8
    with pytest.raises(L.ShodhError):
        S.MiniLMEmbedder(dtype=L.DTYPE_BF16, weights_path=str(tmp_path / "missing.safetensors"))


@pytest.mark.parametrize("export", [False, True])
def test_fused_int8_stages_agree_with_the_round2_kernels(S, tmp_path, export, monkeypatch):
    """SHODH_INT8_STAGES selects, per stage, the fused kernel or the round-2 kernels: the same function, so each stage alone and all
    together must reproduce the all-old embeddings up to the f32 reassociation inside a stage (and the f16-split attention products)."""
    from shodh_memory_amd import _lib as L
    from shodh_memory_amd import embedder as E
    from tests import onnx_writer as W
    kw = {}
    if export:                                                         # non-zero weight zero points: the row-sum terms of every kernel
        cfg, sd, qm = _export_case(E, lambda w: R.quantize_weight_ort(w, per_channel=True), layers=3, vocab=2000, seed=5)
        path = str(tmp_path / "q.onnx")
        W.write_bert(path, sd, cfg.layers, qmodel=qm)
        kw = dict(weights_path=path, layers=cfg.layers, vocab=cfg.vocab)
        vocab = cfg.vocab
    else:
        kw = dict(synthetic_seed=1234)
        vocab = 30522
    ids, mask = _batch(37, 21, vocab)
    keep = mask.sum(1) > 0
    out = {}
    for stages in (0, 1, 2, 4, 8, 15, 47, 68, 111):                   # 68: the pipelined FFN-up kernel feeding the round-2 FFN-down GEMM (which takes the row sums the pass emits); bit 5 (with 0): q|k|v + attention per sequence, quantising the layer input itself; bit 6 (with 2): the FFN-up passes software-pipelined
        monkeypatch.setenv("SHODH_INT8_STAGES", str(stages))
        e8 = S.MiniLMEmbedder(dtype=L.DTYPE_INT8, **kw)
        out[stages] = e8.encode_ids(ids, mask)
        e8.close()
        assert not out[stages][~keep].any()
    for stages in (1, 2, 4, 8, 15, 47, 68, 111):
        c = cos(out[stages][keep], out[0][keep])
        d = np.abs(out[stages] - out[0]).max()
        print("stages", stages, "export", export, "min cosine vs round-2 kernels", c.min(), "max |diff|", d)
        assert c.min() >= 0.99995, (stages, c.min())
    # run-to-run bit identity of the fused path (a race shows up here first); the range keys are order-independent min / max
    monkeypatch.delenv("SHODH_INT8_STAGES")                             # the default: everything fused
    e8 = S.MiniLMEmbedder(dtype=L.DTYPE_INT8, **kw)
    a = e8.encode_ids(ids, mask); b = e8.encode_ids(ids, mask)
    assert a.tobytes() == b.tobytes() == out[111].tobytes()
    # unpadded variant through the fused path too (queries = keys = the real tokens)
    e8u = S.MiniLMEmbedder(dtype=L.DTYPE_INT8, compute_padded=False, **kw)
    monkeypatch.setenv("SHODH_INT8_STAGES", "0")
    e8u0 = S.MiniLMEmbedder(dtype=L.DTYPE_INT8, compute_padded=False, **kw)
    assert cos(e8u.encode_ids(ids, mask)[keep], e8u0.encode_ids(ids, mask)[keep]).min() >= 0.99995
