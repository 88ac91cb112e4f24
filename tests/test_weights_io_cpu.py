"""CPU tests of the model-file readers behind shodh_embedder_load_file (csrc/weights_io.hip, host code): safetensors and ONNX
(fp32 export and dynamic-quantisation export). No device: the host-only view shodh_weight_file_* is compared with
what the files were written from. The files are produced by tests/onnx_writer.py (hand-written protobuf / safetensors containers,
following torch.onnx.export + onnxruntime quantize_dynamic output) and by the `safetensors` package of the image; the real
all-MiniLM-L6-v2 files cannot be fetched offline, so parity with THEM stays unpinned."""
import os

import numpy as np
import pytest

from oracle import int8_ref as R
from tests import onnx_writer as W

f32 = np.float32


@pytest.fixture(scope="module")
def E():
    from shodh_memory_amd import build
    build.build()
    from shodh_memory_amd import embedder
    return embedder


def small_cfg(E, **kw):
    base = dict(vocab=64, hidden=128, layers=2, heads=4, intermediate=256, max_pos=32, type_vocab=2)
    base.update(kw)
    return E.embed_cfg(**base)


def random_sd(E, cfg, seed=0):
    rng = np.random.default_rng(seed)
    blob = (rng.standard_normal(E.param_count(cfg)) * 0.05).astype(f32)
    return blob, E.blob_to_state_dict(blob, cfg)


def test_safetensors_f32_f16_bf16_and_prefix(E, tmp_path):
    cfg = small_cfg(E)
    blob, sd = random_sd(E, cfg)
    p = str(tmp_path / "model.safetensors")
    from safetensors.numpy import save_file                      # the image's own writer: the format as the HF hub ships it
    extra = {"embeddings.position_ids": np.arange(32, dtype=np.int64).reshape(1, -1), "pooler.dense.weight": np.zeros((128, 128), f32),
             "pooler.dense.bias": np.zeros(128, f32)}
    save_file({**{k: np.ascontiguousarray(v) for k, v in sd.items()}, **extra}, p)
    wf = E.WeightFile(p, cfg)
    assert wf.blob().tobytes() == blob.tobytes()
    assert wf.quantized("encoder.layer.0.attention.self.query.weight", (128, 128)) is None
    # hand-written containers: module prefix, F16, BF16
    for dtype, tol in (("F32", 0.0), ("F16", 2e-4), ("BF16", 2e-3)):
        q = str(tmp_path / ("m_%s.safetensors" % dtype))
        W.write_safetensors(q, sd, dtype=dtype, prefix="0.auto_model.")
        got = E.WeightFile(q, cfg).blob()
        assert np.abs(got - blob).max() <= tol * max(1.0, float(np.abs(blob).max()))
        if dtype == "F16":
            assert got.tobytes() == blob.astype(np.float16).astype(f32).tobytes()


def test_safetensors_errors(E, tmp_path):
    from shodh_memory_amd import _lib as L
    cfg = small_cfg(E)
    blob, sd = random_sd(E, cfg)
    missing = dict(sd)
    del missing["encoder.layer.1.output.LayerNorm.bias"]
    p = str(tmp_path / "missing.safetensors")
    W.write_safetensors(p, missing)
    with pytest.raises(L.ShodhError) as e:
        E.WeightFile(p, cfg)
    assert "encoder.layer.1.output.LayerNorm.bias" in str(e.value)
    bad = dict(sd)
    bad["embeddings.LayerNorm.weight"] = np.zeros(7, f32)
    p = str(tmp_path / "shape.safetensors")
    W.write_safetensors(p, bad)
    with pytest.raises(L.ShodhError):
        E.WeightFile(p, cfg)
    p = str(tmp_path / "trunc.safetensors")
    open(p, "wb").write(b"\x10\x00\x00")
    with pytest.raises(L.ShodhError):
        E.WeightFile(p, cfg)
    with pytest.raises(L.ShodhError):
        E.WeightFile(str(tmp_path / "nope.safetensors"), cfg)


@pytest.mark.parametrize("raw,named_bias,packed_dims", [(True, True, False), (False, True, True), (True, False, False)])
def test_onnx_fp32_export(E, tmp_path, raw, named_bias, packed_dims):
    cfg = small_cfg(E)
    blob, sd = random_sd(E, cfg, seed=1)
    p = str(tmp_path / "model.onnx")
    W.write_bert(p, sd, 2, qmodel=None, raw=raw, named_bias=named_bias, packed_dims=packed_dims)
    wf = E.WeightFile(p, cfg)
    assert wf.blob().tobytes() == blob.tobytes()                       # anonymous [K][N] MatMul constants found by their bias (or by graph order) and transposed back
    assert wf.quantized("encoder.layer.1.intermediate.dense.weight", (256, 128)) is None


RULES = {
    "u8_per_tensor_asym": lambda w: R.quantize_weight_ort(w),
    "u8_per_channel_asym": lambda w: R.quantize_weight_ort(w, per_channel=True),
    "u8_symmetric": lambda w: R.quantize_weight_ort(w, symmetric=True),
    "u8_per_channel_7bit": lambda w: R.quantize_weight_ort(w, per_channel=True, reduce_range=True),
    "s8_symmetric": lambda w: R.quantize_weight_ort(w, symmetric=True, signed=True),
    "s8_per_channel_asym": lambda w: R.quantize_weight_ort(w, per_channel=True, signed=True),
}


@pytest.mark.parametrize("rule", sorted(RULES))
@pytest.mark.parametrize("raw", [True, False])
def test_onnx_dynamic_quantised_export(E, tmp_path, rule, raw):
    cfg = small_cfg(E)
    blob, sd = random_sd(E, cfg, seed=2)
    word_rule = (lambda w: R.quantize_weight_ort(w, signed=rule.startswith("s8"))) if "per_channel" in rule else RULES[rule]      # Gather tables: one scale
    qm = R.quantize_model(sd, 2, rule=RULES[rule], word_rule=word_rule)
    p = str(tmp_path / "model_quantized.onnx")
    W.write_bert(p, sd, 2, qmodel=qm, raw=raw)
    wf = E.WeightFile(p, cfg)
    got = E.blob_to_state_dict(wf.blob(), cfg)
    for name, a in sd.items():
        if name in qm:
            q, sc, zp = qm[name]
            assert got[name].tobytes() == R.dequantize(q, sc, zp).tobytes(), name         # DequantizeLinear of the file's own tensor
            qs, ss, zs = wf.quantized(name, q.shape)
            off = 0 if q.dtype == np.int8 else 128                                            # the library's storage: signed bytes, zero point in the same terms
            assert np.array_equal(qs.astype(np.int32), q.astype(np.int32) - off), name
            assert np.array_equal(zs, zp.astype(np.int32) - off) and ss.tobytes() == sc.tobytes(), name
        else:
            assert got[name].tobytes() == a.tobytes(), name
            assert wf.quantized(name, a.shape if a.ndim == 2 else (1, a.size)) is None if a.ndim == 2 else True


def test_onnx_errors(E, tmp_path):
    from shodh_memory_amd import _lib as L
    cfg = small_cfg(E)
    blob, sd = random_sd(E, cfg, seed=3)
    p = str(tmp_path / "garbage.onnx")
    open(p, "wb").write(os.urandom(300))
    with pytest.raises(L.ShodhError):
        E.WeightFile(p, cfg)
    # a model with the wrong width
    other = small_cfg(E, hidden=256, heads=8)
    _, sd2 = random_sd(E, other, seed=4)
    p = str(tmp_path / "wide.onnx")
    W.write_bert(p, sd2, 2)
    with pytest.raises(L.ShodhError):
        E.WeightFile(p, cfg)


def test_tensor_by_tensor_handover_needs_a_device_only_at_finish(E):
    """shodh_embedder_load_tensor / _load_quantized validate names and shapes on the host; exercised on the GPU in test_encoder_int8_gpu.py.
    Here: the symbol table and the constants."""
    from shodh_memory_amd import _lib as L
    assert (L.WEIGHT_ABSENT, L.WEIGHT_F32, L.WEIGHT_EXPORT_Q8, L.WEIGHT_SELF_Q8) == (0, 1, 2, 3)
    for s in ("shodh_embedder_load_file", "shodh_embedder_load_tensor", "shodh_embedder_load_quantized", "shodh_embedder_finish_weights", "shodh_embedder_weight_source"):
        assert hasattr(L.lib(), s)


def test_crafted_files_are_rejected_not_fatal(E, tmp_path):
    """a file is untrusted input. Headers whose numbers are negative, fractional, overflowing or run to the end of the header without a
    terminator, metadata nested beyond any sane depth, shapes whose product wraps, ONNX dims that would size a huge allocation: every one is an
    error status through the C ABI, none aborts the process or reads past the header."""
    import json
    import struct
    from shodh_memory_amd import _lib as L
    cfg = small_cfg(E)
    blob, sd = random_sd(E, cfg)

    def st_file(name, header_text, payload=b"\0" * 64):
        h = header_text.encode()
        p = str(tmp_path / name)
        open(p, "wb").write(struct.pack("<Q", len(h)) + h + payload)
        return p
    key = "embeddings.LayerNorm.weight"
    cases = {
        "neg.safetensors": '{"%s":{"dtype":"F32","shape":[-128],"data_offsets":[0,512]}}' % key,
        "frac.safetensors": '{"%s":{"dtype":"F32","shape":[128],"data_offsets":[0,5.12e2]}}' % key,
        "huge.safetensors": '{"%s":{"dtype":"F32","shape":[99999999999999999999999],"data_offsets":[0,512]}}' % key,
        "wrap.safetensors": '{"%s":{"dtype":"F32","shape":[4294967296,4294967296,128],"data_offsets":[0,512]}}' % key,
        "deep.safetensors": '{"__metadata__":' + "[" * 5000 + "]" * 5000 + "}",
        "open.safetensors": '{"%s":{"dtype":"F32","shape":[128],"data_offsets":[0,51' % key,          # the header ends inside a number: nothing behind it but the payload
    }
    for name, text in cases.items():
        with pytest.raises(L.ShodhError):
            E.WeightFile(st_file(name, text, payload=b"9" * 4096), cfg)
    # ONNX: a scale tensor whose dims promise 2^40 elements (and a weight whose dims are negative) next to a few bytes of payload
    good = str(tmp_path / "good.onnx")
    qm = R.quantize_model(sd, cfg.layers, rule=lambda w: R.quantize_weight_ort(w, per_channel=True), word_rule=lambda w: R.quantize_weight_ort(w))
    W.write_bert(good, sd, cfg.layers, qmodel=qm)
    assert E.WeightFile(good, cfg).blob().size == blob.size
    for bad_dims in ([1 << 40], [1 << 62, 1 << 62], [-5]):
        t = W.tensor("x_scale", np.zeros(4, f32))
        # rebuild the tensor proto with crafted dims: field 1 (dims, varint) repeated, then the rest of the original tensor minus its own dims
        body = b"".join(W._vi(1, d & ((1 << 64) - 1)) for d in bad_dims) + W._vi(2, 1) + W._ld(8, b"encoder.layer.0.output.dense.weight_scale") + W._ld(9, b"\0" * 16)
        p = str(tmp_path / ("dims_%d.onnx" % (len(bad_dims) * 7 + (bad_dims[0] & 7))))
        raw = open(good, "rb").read()
        # append one more initializer to the graph: model = field 7 (graph) -> graph field 5 (initializer). Appending a second graph field merges in protobuf.
        open(p, "wb").write(raw + W._ld(7, W._ld(5, body)))
        try:
            wf = E.WeightFile(p, cfg)          # either the crafted tensor is ignored (never matched) ...
            assert wf.blob().size == blob.size
        except L.ShodhError:
            pass                               # ... or the file is rejected: both are fine, a crash is not
        del t
