"""GPU tests of SHODH_QUANT_SCOPE_PER_TEXT (round 4): the INT8 function the reference's hot path computes.

`remember`, `index_memory` and `recall` call `encode()` text by text (memory/mod.rs:1037, retrieval.rs:673, :708, :878): one
`session.run` on `[1, max_len]` (minilm.rs:883-982), so every DynamicQuantizeLinear range spans ONE text's padded tensor. With
quant_scope = PER_TEXT a batch of N texts is that function N times:
  * byte-equal to N calls with one text each (and to the batch scope run on one text: a batch of one has one range per tensor either way),
  * equal to the numpy restatement called row by row (oracle/int8_ref.py::encode(per_text=True)) within its float tolerance,
  * independent of the batch mates and of the position in the batch,
and shapes the per-sequence kernels do not take (unpadded tensors, max_len not a multiple of 128) run one text per forward.
Parity with model_quint8_avx2.onnx itself stays UNPINNED (no ONNX Runtime / checkpoint offline)."""
import os

import numpy as np
import pytest

from oracle import int8_ref as R
from tests.test_encoder_int8_gpu import _batch, _export_case, cos

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
f32 = np.float32


@pytest.fixture(scope="module")
def S():
    from shodh_memory_amd import build
    build.build()
    import shodh_memory_amd as s
    return s


def _embedder(S, tmp_path, export, **extra):
    from shodh_memory_amd import _lib as L
    from shodh_memory_amd import embedder as E
    from tests import onnx_writer as W
    if export:                                                         # uint8 weights with per-channel zero points: the row-sum terms of every kernel
        cfg, sd, qm = _export_case(E, lambda w: R.quantize_weight_ort(w, per_channel=True), layers=2, vocab=2000, seed=11)
        path = str(tmp_path / "q.onnx")
        if not os.path.exists(path):
            W.write_bert(path, sd, cfg.layers, qmodel=qm)
        return S.MiniLMEmbedder(dtype=L.DTYPE_INT8, weights_path=path, layers=cfg.layers, vocab=cfg.vocab, **extra), sd, qm, cfg.layers, cfg.vocab
    sd = E.blob_to_state_dict(E.synthetic_weights(1234))
    return S.MiniLMEmbedder(synthetic_seed=1234, dtype=L.DTYPE_INT8, **extra), sd, None, 6, 30522


@pytest.mark.parametrize("export", [False, True])
def test_batch_of_n_is_n_calls_of_one(S, tmp_path, export):
    from shodh_memory_amd import _lib as L
    e8, sd, qm, layers, vocab = _embedder(S, tmp_path, export)
    ids, mask = _batch(64, 31, vocab)
    keep = mask.sum(1) > 0
    assert e8.quant_scope() == L.QUANT_SCOPE_BATCH                    # the C default: encode_ids == the reference's encode_batch
    together = e8.encode_ids(ids, mask, scope=L.QUANT_SCOPE_PER_TEXT)
    assert e8.quant_scope() == L.QUANT_SCOPE_BATCH                    # restored
    assert not together[~keep].any()
    one_by_one = np.concatenate([e8.encode_ids(ids[i:i + 1], mask[i:i + 1], scope=L.QUANT_SCOPE_PER_TEXT) for i in range(64)], 0)
    assert together.tobytes() == one_by_one.tobytes()                 # N texts == N x encode(), bit for bit
    # a batch of ONE text is the same function in either scope (one range per tensor), and the batch-scope kernels are the round-3 ones
    batch_scope_single = np.concatenate([e8.encode_ids(ids[i:i + 1], mask[i:i + 1]) for i in range(0, 64, 7)], 0)
    c = cos(together[::7][keep[::7]], batch_scope_single[keep[::7]])
    d = np.abs(together[::7] - batch_scope_single).max()
    print("per-sequence kernels vs batch kernels on one text: min cosine", c.min(), "max |diff|", d)
    assert c.min() >= 0.99995
    # independent of batch mates and position: another batch holding some of the same texts
    perm = np.array([5, 63, 0, 17, 40, 1])
    other = e8.encode_ids(ids[perm], mask[perm], scope=L.QUANT_SCOPE_PER_TEXT)
    assert other.tobytes() == together[perm].tobytes()
    # ... which the batch scope is NOT (the reference's encode_batch: ranges over all rows)
    batch_scope = e8.encode_ids(ids, mask)
    assert batch_scope.tobytes() != together.tobytes() and cos(batch_scope[keep], together[keep]).min() >= 0.97
    # run to run
    assert e8.encode_ids(ids, mask, scope=L.QUANT_SCOPE_PER_TEXT).tobytes() == together.tobytes()
    # against the restatement called row by row
    rows = np.array([0, 1, 3, 63])
    exp = R.encode(sd, ids[rows], mask[rows], layers=layers, qmodel=qm, per_text=True)
    c = cos(together[rows], exp)
    print("export" if export else "self-quantised", "per-text cosine vs restatement:", c)
    assert c.min() >= 0.9999 and np.abs(together[rows] - exp).max() < 2e-3
    # the handle's default scope can be PER_TEXT as well (cfg.quant_scope / shodh_embedder_set_quant_scope)
    e8.set_quant_scope(L.QUANT_SCOPE_PER_TEXT)
    assert e8.encode_ids(ids, mask).tobytes() == together.tobytes()
    with pytest.raises(L.ShodhError):
        e8.set_quant_scope(7)


def test_per_text_device_api_and_large_batch(S, tmp_path):
    """device pointers, a batch above the per-forward split (4096 texts), and cfg.quant_scope at creation"""
    import torch
    from shodh_memory_amd import _lib as L
    e8, _, _, _, vocab = _embedder(S, tmp_path, False, quant_scope=L.QUANT_SCOPE_PER_TEXT)
    assert e8.quant_scope() == L.QUANT_SCOPE_PER_TEXT
    ids, mask = _batch(4200, 77, vocab)
    d_ids = torch.from_numpy(ids).cuda(); d_mask = torch.from_numpy(mask).cuda()
    out = e8.encode_ids_device(d_ids, d_mask)
    torch.cuda.synchronize()
    out = out.cpu().numpy()
    pick = np.array([0, 1, 2, 4095, 4096, 4199])
    host = e8.encode_ids(ids[pick], mask[pick])
    assert out[pick].tobytes() == host.tobytes()
    keep = mask.sum(1) > 0
    assert np.allclose(np.linalg.norm(out[keep], axis=1), 1, atol=1e-4) and not out[~keep].any()


@pytest.mark.parametrize("kw", [dict(compute_padded=False), dict(max_length=64), dict(max_length=128)])
def test_per_text_on_shapes_without_the_per_sequence_kernels(S, tmp_path, kw):
    """unpadded tensors / max_len 64 run one text per forward (same function by definition); max_len 128 takes the per-sequence kernels"""
    from shodh_memory_amd import _lib as L
    ML = kw.get("max_length", 256)
    e8, sd, _, layers, vocab = _embedder(S, tmp_path, False, **kw)
    ids, mask = _batch(9, 3, vocab)
    ids, mask = ids[:, :ML].copy(), mask[:, :ML].copy()
    together = e8.encode_ids(ids, mask, scope=L.QUANT_SCOPE_PER_TEXT)
    singles = np.concatenate([e8.encode_ids(ids[i:i + 1], mask[i:i + 1]) for i in range(9)], 0)
    if ML == 128:
        assert cos(together[mask.sum(1) > 0], singles[mask.sum(1) > 0]).min() >= 0.99995
        again = np.concatenate([e8.encode_ids(ids[i:i + 1], mask[i:i + 1], scope=L.QUANT_SCOPE_PER_TEXT) for i in range(9)], 0)
        assert together.tobytes() == again.tobytes()
    else:
        assert together.tobytes() == singles.tobytes()
    if kw.get("compute_padded", True):
        exp = R.encode(sd, ids[:2], mask[:2], layers=layers, per_text=True)
        assert cos(together[:2], exp).min() >= 0.9999


def test_embedder_trait_scopes(S, tmp_path):
    """Embedder.encode_batch = the reference's batch call; encode_each = N x encode()"""
    tokenizers = pytest.importorskip("tokenizers")
    from tokenizers import Tokenizer, models, pre_tokenizers, processors
    from shodh_memory_amd import _lib as L
    words = ["[PAD]", "[UNK]", "[CLS]", "[SEP]"] + ["w%d" % i for i in range(200)]
    tok = Tokenizer(models.WordLevel({w: i for i, w in enumerate(words)}, unk_token="[UNK]"))
    tok.pre_tokenizer = pre_tokenizers.Whitespace()
    tok.post_processor = processors.TemplateProcessing(single="[CLS] $A [SEP]", special_tokens=[("[CLS]", 2), ("[SEP]", 3)])
    e8 = S.MiniLMEmbedder(tokenizer=tok, synthetic_seed=1234, dtype=L.DTYPE_INT8)
    texts = ["w1 w2 w3", "", "w9 " * 40, "w100 w7"]
    each = e8.encode_each(texts)
    for t, v in zip(texts, each):
        assert v.tobytes() == e8.encode(t).tobytes()
    batch = e8.encode_batch(texts)
    assert not batch[1].any() and not each[1].any()
    assert any(a.tobytes() != b.tobytes() for a, b in zip(batch, each))      # the batch call's ranges span all rows


def test_first_contact_verifier_end_to_end(S, tmp_path, monkeypatch, capsys):
    """tools/verify_real_model.py on a directory laid out like the HuggingFace repository (files written by tests/onnx_writer.py: the real ones
    cannot be fetched here): checksums differ from the pinned ones by construction, the reader's tensor map and every GPU check run and pass."""
    import importlib.util
    import json
    from shodh_memory_amd import embedder as E
    from tests import onnx_writer as W
    spec = importlib.util.spec_from_file_location("verify_real_model", os.path.join(ROOT, "tools", "verify_real_model.py"))
    V = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(V)
    cfg, sd, qm = _export_case(E, lambda w: R.quantize_weight_ort(w, per_channel=True), layers=2, vocab=30522, seed=3)
    os.makedirs(tmp_path / "onnx")
    W.write_bert(str(tmp_path / "onnx" / "model_quint8_avx2.onnx"), sd, cfg.layers, qmodel=qm)
    W.write_bert(str(tmp_path / "onnx" / "model.onnx"), sd, cfg.layers)
    monkeypatch.setenv("SHODH_VERIFY_CFG", json.dumps(dict(layers=2))); monkeypatch.setenv("SHODH_VERIFY_JSON", str(tmp_path / "rep.json"))
    rc = V.main(["verify_real_model.py", str(tmp_path)])
    out = capsys.readouterr().out
    print(out)
    rep = json.load(open(tmp_path / "rep.json"))
    g = rep["steps"]["gpu"]
    assert rc == 0 and g["ok"], g
    assert g["fp32_pad128_vs_pad256_bit_identical"] and g["int8_batch_equals_n_calls"] and g["int8_weight_source"].startswith("the export")
    assert g["full_unit_norm_max_dev"] < 1e-5 and g["quantized_unit_norm_max_dev"] < 1e-5
    assert rep["steps"]["sha256"]["quantized"]["status"].startswith("DIFFERS")


@pytest.mark.parametrize("export", [False, True])
def test_per_sequence_tail_kernel_equals_the_separate_kernels(S, tmp_path, monkeypatch, export):
    """SHODH_INT8_STAGES bit 7: attention output + LayerNorm + both quantising passes as ONE kernel per sequence (attn_out_ln_quant_seq_kernel) or as
    the four launches with per-sequence range slots (act_quant_seq, i8_stream_kernel<RESID_LN, PS>, act_quant_seq). Same sums in the same order:
    the same bits."""
    from shodh_memory_amd import _lib as L
    out = {}
    for stages in ("0x6F", "0xEF"):
        monkeypatch.setenv("SHODH_INT8_STAGES", stages)
        e8, _, _, _, vocab = _embedder(S, tmp_path, export, quant_scope=L.QUANT_SCOPE_PER_TEXT)
        ids, mask = _batch(40, 13, vocab)
        out[stages] = e8.encode_ids(ids, mask)
        e8.close()
    assert out["0x6F"].tobytes() == out["0xEF"].tobytes()


@pytest.mark.parametrize("scope", ["batch", "per_text"])
def test_zero_point_terms_in_float_arithmetic_are_the_same_bits(S, tmp_path, monkeypatch, scope):
    """SHODH_INT8_STAGES bit 8: with an export's weight zero points the FFN-up passes form float(acc) + corr * rsz - zw * rowsum_a in float arithmetic
    (i8_stream_gelu_kernel<., 2, .>) where the tensor's own bytes prove every partial sum an integer below 2^24 (QWeight::zw_bound), instead of an
    integer multiply + subtract + add per value before the conversion. Exact either way: the same embeddings, byte for byte."""
    from shodh_memory_amd import _lib as L
    out = {}
    for stages in ("0xEF", "0x1EF"):
        monkeypatch.setenv("SHODH_INT8_STAGES", stages)
        e8, _, _, _, vocab = _embedder(S, tmp_path, True, quant_scope=L.QUANT_SCOPE_PER_TEXT if scope == "per_text" else L.QUANT_SCOPE_BATCH)
        ids, mask = _batch(48, 17, vocab)
        out[stages] = e8.encode_ids(ids, mask)
        e8.close()
    assert out["0xEF"].tobytes() == out["0x1EF"].tobytes()
