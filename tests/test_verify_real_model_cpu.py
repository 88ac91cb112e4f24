"""tools/verify_real_model.py (the first-contact verifier for the real all-MiniLM-L6-v2 files) must run offline: without files it says
"nothing to verify"; on a dynamic-quantisation export written by tests/onnx_writer.py it reports a checksum that DIFFERS from the pinned
one (downloader.rs:38-53) and reads the tensor map -- per channel / per tensor, zero points, bit width -- back correctly. No device."""
import importlib.util
import json
import os

import numpy as np
import pytest

from oracle import int8_ref as R
from tests import onnx_writer as W

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
f32 = np.float32


@pytest.fixture(scope="module")
def V():
    from shodh_memory_amd import build
    build.build()
    spec = importlib.util.spec_from_file_location("verify_real_model", os.path.join(ROOT, "tools", "verify_real_model.py"))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    return m


def test_no_files_is_not_an_error(V, tmp_path, capsys):
    assert V.main(["verify_real_model.py", str(tmp_path)]) == 0
    assert "nothing to verify" in capsys.readouterr().out
    assert V.main(["verify_real_model.py"]) == 0


def test_pinned_checksums_are_the_reference_literals(V):
    # downloader.rs:38-53
    assert V.PINNED["quantized"][0] == "b941bf19f1f1283680f449fa6a7336bb5600bdcd5f84d10ddc5cd72218a0fd21"
    assert V.PINNED["full"][0] == "6fd5d72fe4589f189f8ebc006442dbb529bb7ce38f8082112682524616046452"
    assert V.PINNED["tokenizer"][0] == "be50c3628f2bf5bb5e3a7f17b1f74611b2561a3a27eeab05e5aa30f411572037"


@pytest.mark.parametrize("rule,want", [("per_channel_asym", ("per_channel", False, 8)), ("per_tensor_sym_s8", ("per_tensor", True, 8)), ("reduce_range", ("per_channel", False, 7))])
def test_tensor_map_of_a_written_export(V, tmp_path, monkeypatch, capsys, rule, want):
    from shodh_memory_amd import embedder as E
    cfg_kw = dict(vocab=64, hidden=128, layers=2, heads=4, intermediate=256, max_pos=32, type_vocab=2)
    cfg = E.embed_cfg(**cfg_kw)
    rng = np.random.default_rng(1)
    sd = E.blob_to_state_dict((rng.standard_normal(E.param_count(cfg)) * 0.05).astype(f32) + f32(0.003), cfg)
    rules = {"per_channel_asym": lambda w: R.quantize_weight_ort(w, per_channel=True), "per_tensor_sym_s8": lambda w: R.quantize_weight_ort(w, symmetric=True, signed=True),
             "reduce_range": lambda w: R.quantize_weight_ort(w, per_channel=True, reduce_range=True)}
    qm = R.quantize_model(sd, cfg.layers, rule=rules[rule], word_rule=lambda w: R.quantize_weight_ort(w))
    os.makedirs(tmp_path / "onnx")
    W.write_bert(str(tmp_path / "onnx" / "model_quint8_avx2.onnx"), sd, cfg.layers, qmodel=qm)
    W.write_bert(str(tmp_path / "onnx" / "model.onnx"), sd, cfg.layers)
    files = V.find_files(str(tmp_path))
    assert set(files) == {"quantized", "full"}
    h = V.check_hashes(files)
    assert h["quantized"]["status"].startswith("DIFFERS") and len(h["quantized"]["sha256"]) == 64
    d = V.describe_weights(files["quantized"], cfg_kw)
    t = d["tensors"]["encoder.layer.1.intermediate.dense.weight"]
    assert (t["granularity"], t["symmetric"], t["bits"]) == want and t["stored"] == "q8"
    assert d["tensors"]["embeddings.word_embeddings.weight"]["granularity"] == "per_tensor" and d["finite"]
    full = V.describe_weights(files["full"], cfg_kw)
    assert all(v["stored"] == "f32" for v in full["tensors"].values())
    # the command line on the same directory (no GPU here: steps 1 and 2 only), with the JSON report
    monkeypatch.setenv("SHODH_VERIFY_CFG", json.dumps(cfg_kw)); monkeypatch.setenv("SHODH_VERIFY_JSON", str(tmp_path / "rep.json"))
    import torch
    if not torch.cuda.is_available():
        assert V.main(["verify_real_model.py", str(tmp_path)]) == 0
        out = capsys.readouterr().out
        assert "[sha256] quantized" in out and "[reader] quantized" in out and "[gpu] skipped" in out
        rep = json.load(open(tmp_path / "rep.json"))
        assert rep["steps"]["weights"]["quantized"]["tensors"]["encoder.layer.0.output.dense.weight"]["stored"] == "q8"
