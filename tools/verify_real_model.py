#!/usr/bin/env python3
"""First-contact verifier for the REAL all-MiniLM-L6-v2 files.

Parity of the HIP encoder with the reference's model files is UNPINNED in this repository: the reference downloads them at first
run (src/embeddings/downloader.rs:29-53) and there is no network here. This script turns "unpinned" into "pinned within one command"
on the day the files are supplied:

    python tools/verify_real_model.py <dir>        # <dir> holds onnx/model_quint8_avx2.onnx (or model_quantized.onnx), onnx/model.onnx,
                                                   # tokenizer.json, optionally model.safetensors -- the HuggingFace repository layout

  1. sha256 of every file found against the checksums the reference pins (downloader.rs:38-53, commit c9745ed1 of
     sentence-transformers/all-MiniLM-L6-v2; the literals are restated below);
  2. what the library's reader makes of each weight file (shodh_weight_file_*: no device needed): which tensors arrive quantised, per tensor
     or per output channel, symmetric or with zero points, 8 or 7 bit -- the scheme `model_quint8_avx2.onnx` REALLY uses, which cannot be
     known offline;
  3. on a GPU, the numeric facts the reference states about these files:
       - unit norm of every embedding, |norm - 1| < 1e-5                                   (minilm.rs:1416-1417)
       - fp32 export: padding to 128 vs 256 positions gives the same pooled vector to the last bit     (minilm.rs:153-154)
       - quint8 export: pad-128 vs pad-256 worst cosine 0.9859 on the reference's texts    (minilm.rs:591, :1390) -- ours is reported, gate >= 0.975
       - HIP INT8 (quant_scope PER_TEXT = encode()) vs HIP fp32 of the full export: per-text cosine (expected >= 0.98)
       - batch of N (PER_TEXT) == N x encode(), bit for bit

Without files it says so and exits 0 (it must run offline); tests/test_verify_real_model_cpu.py drives steps 1-2 on a file written by
tests/onnx_writer.py. Exit status 1 = a file is present and a check failed."""
import hashlib
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

# downloader.rs:38-53 (ModelChecksums), pinned to sentence-transformers/all-MiniLM-L6-v2 @ c9745ed1d9f207416be6d2e6f8de32d1f16199bf
PINNED = {
    "quantized": ("b941bf19f1f1283680f449fa6a7336bb5600bdcd5f84d10ddc5cd72218a0fd21", ("onnx/model_quint8_avx2.onnx", "model_quint8_avx2.onnx", "model_quantized.onnx", "onnx/model_quantized.onnx")),
    "full": ("6fd5d72fe4589f189f8ebc006442dbb529bb7ce38f8082112682524616046452", ("onnx/model.onnx", "model.onnx")),
    "tokenizer": ("be50c3628f2bf5bb5e3a7f17b1f74611b2561a3a27eeab05e5aa30f411572037", ("tokenizer.json",)),
    "safetensors": (None, ("model.safetensors",)),          # the checkpoint itself: the reference never downloads it, no pinned checksum
}
QUANTISED_SUFFIXES = ("attention.self.query.weight", "attention.self.key.weight", "attention.self.value.weight", "attention.output.dense.weight",
                      "intermediate.dense.weight", "output.dense.weight")
SAMPLE_TEXTS = [
    "The quick brown fox jumps over the lazy dog.",
    "Remember to rotate the API keys before the quarterly audit.",
    "Rust's borrow checker rejected the second mutable reference.",
    "Meeting notes: the vector index rebuild finished in 41 minutes.",
    "short",
    "A considerably longer passage about memory systems: episodic traces decay unless they are rehearsed, semantic facts are consolidated "
    "from repeated episodes, and retrieval is a cue-dependent reconstruction whose success depends on the overlap between the cue and the trace. " * 3,
]


def sha256(path):
    h = hashlib.sha256()
    with open(path, "rb") as f:
        for blk in iter(lambda: f.read(1 << 20), b""):
            h.update(blk)
    return h.hexdigest()


def find_files(d):
    out = {}
    for key, (_, rels) in PINNED.items():
        for rel in rels:
            p = os.path.join(d, rel)
            if os.path.isfile(p):
                out[key] = p
                break
    return out


def check_hashes(files):
    rep = {}
    for key, path in files.items():
        want = PINNED[key][0]
        got = sha256(path)
        rep[key] = dict(path=path, sha256=got, pinned=want, status="no pinned checksum" if want is None else ("OK" if got == want else "DIFFERS from the file the reference pins"))
    return rep


def describe_weights(path, cfg_kw=None):
    """What shodh_weight_file_* reads from a weight file: per quantisable tensor its scheme. Host only."""
    import numpy as np
    from shodh_memory_amd import embedder as E
    cfg = E.embed_cfg(**(cfg_kw or {}))
    wf = E.WeightFile(path, cfg)
    H, I = cfg.hidden, cfg.intermediate
    names = [("embeddings.word_embeddings.weight", (cfg.vocab, H))]
    for l in range(cfg.layers):
        p = "encoder.layer.%d." % l
        for sfx in QUANTISED_SUFFIXES:
            shape = (I, H) if sfx == "intermediate.dense.weight" else ((H, I) if sfx == "output.dense.weight" else (H, H))
            names.append((p + sfx, shape))
    tensors = {}
    for name, shape in names:
        q = wf.quantized(name, shape)
        if q is None:
            tensors[name] = dict(stored="f32")
            continue
        qb, sc, zp = q                                                  # signed storage (uint8 sources minus 128), zero points in the same terms
        lo, hi = int(qb.min()), int(qb.max())
        # in the stored (signed) terms: a uint8 tensor with zero point z arrives as bytes - 128 with zero point z - 128
        span = hi - lo
        tensors[name] = dict(stored="q8", granularity="per_channel" if sc.size > 1 else "per_tensor", n_scale=int(sc.size),
                             zero_point_min=int(zp.min()), zero_point_max=int(zp.max()),
                             symmetric=bool((zp == 0).all()),                 # (0 in the stored terms: int8 zero point 0, uint8 zero point 128)
                             byte_min=lo, byte_max=hi, bits=7 if span <= 127 else 8,
                             scale_min=float(sc.min()), scale_max=float(sc.max()))
    blob = wf.blob()
    wf.close()
    schemes = sorted({(t.get("stored"), t.get("granularity"), t.get("symmetric"), t.get("bits")) for t in tensors.values()}, key=str)
    return dict(path=path, n_params=int(blob.size), finite=bool(np.isfinite(blob).all()), tensors=tensors, schemes=[list(s) for s in schemes])


def _tokenise(tok_path, texts, max_len):
    import numpy as np
    ids = np.zeros((len(texts), max_len), np.int32); mask = np.zeros((len(texts), max_len), np.uint8)
    if tok_path:
        from tokenizers import Tokenizer
        tok = Tokenizer.from_file(tok_path)
        tok.enable_truncation(max_length=128); tok.no_padding()          # minilm.rs:112-117
        for i, enc in enumerate(tok.encode_batch(list(texts), add_special_tokens=True)):
            n = min(len(enc.ids), max_len)
            ids[i, :n] = enc.ids[:n]; mask[i, :n] = enc.attention_mask[:n]
    else:                                                                # no tokenizer.json: synthetic ids of the same lengths (the checks below do not need real words)
        rng = np.random.default_rng(0)
        for i, t in enumerate(texts):
            n = max(3, min(128, len(t.split()) + 2))
            ids[i, :n] = rng.integers(1000, 29000, n); ids[i, 0] = 101; ids[i, n - 1] = 102; mask[i, :n] = 1
    return ids, mask


def gpu_checks(files, cfg_kw=None):
    import numpy as np
    import shodh_memory_amd as S
    from shodh_memory_amd import _lib as L
    kw = dict(cfg_kw or {})
    rep, ok = {}, True

    def cos(a, b):
        return (a * b).sum(1) / np.maximum(np.linalg.norm(a, axis=1) * np.linalg.norm(b, axis=1), 1e-30)
    tokp = files.get("tokenizer")
    emb = {}
    for key, dtype, scope in (("full", L.DTYPE_FP32, None), ("quantized", L.DTYPE_INT8, L.QUANT_SCOPE_PER_TEXT)):
        if key not in files:
            continue
        for ml in (256, 128):
            e = S.MiniLMEmbedder(dtype=dtype, weights_path=files[key], max_length=ml, **({"quant_scope": scope} if scope is not None else {}), **kw)
            ids, mask = _tokenise(tokp, SAMPLE_TEXTS, ml)
            emb[(key, ml)] = e.encode_ids(ids, mask)
            if key == "quantized" and ml == 256:
                one = np.concatenate([e.encode_ids(ids[i:i + 1], mask[i:i + 1]) for i in range(len(SAMPLE_TEXTS))], 0)
                rep["int8_batch_equals_n_calls"] = bool(one.tobytes() == emb[(key, ml)].tobytes()); ok &= rep["int8_batch_equals_n_calls"]
                src = e.weight_source("encoder.layer.0.output.dense.weight")
                rep["int8_weight_source"] = {L.WEIGHT_EXPORT_Q8: "the export's own bytes", L.WEIGHT_SELF_Q8: "floats quantised by this library (FALLBACK)", L.WEIGHT_F32: "f32"}.get(src, str(src))
            e.close()
        norms = np.linalg.norm(emb[(key, 256)], axis=1)
        rep[key + "_unit_norm_max_dev"] = float(np.abs(norms - 1).max()); ok &= rep[key + "_unit_norm_max_dev"] < 1e-5      # minilm.rs:1416-1417
    if "full" in files:
        rep["fp32_pad128_vs_pad256_bit_identical"] = bool(emb[("full", 128)].tobytes() == emb[("full", 256)].tobytes())         # minilm.rs:153-154
        rep["fp32_pad128_vs_pad256_max_abs_diff"] = float(np.abs(emb[("full", 128)] - emb[("full", 256)]).max())
        ok &= rep["fp32_pad128_vs_pad256_bit_identical"]
    if "quantized" in files:
        c = cos(emb[("quantized", 128)], emb[("quantized", 256)])
        rep["quint8_pad128_vs_pad256_worst_cosine"] = float(c.min()); rep["reference_figure_minilm_rs_591"] = 0.9859
        ok &= c.min() >= 0.975
    if "quantized" in files and "full" in files:
        c = cos(emb[("quantized", 256)], emb[("full", 256)])
        rep["int8_per_text_vs_fp32_cosine"] = [float(x) for x in c]
        ok &= c.min() >= 0.98
    rep["ok"] = bool(ok)
    return rep


def main(argv):
    d = argv[1] if len(argv) > 1 else os.environ.get("SHODH_MODEL_DIR", "")
    out = dict(dir=d, steps={})
    files = find_files(d) if d and os.path.isdir(d) else {}
    if not files:
        print("no model files found%s: nothing to verify (parity with the real checkpoint stays UNPINNED).\n"
              "expected under the directory: %s" % ((" under " + d) if d else "", ", ".join(r[0] for _, r in PINNED.values())))
        return 0
    failed = False
    cfg_kw = json.loads(os.environ.get("SHODH_VERIFY_CFG", "{}"))       # other BERT shapes (the tests' small files): {"layers": 2, "vocab": 3000, ...}
    out["steps"]["sha256"] = check_hashes(files)
    for key, r in out["steps"]["sha256"].items():
        print("[sha256] %-11s %s  %s" % (key, r["sha256"][:16] + "...", r["status"]))
    out["steps"]["weights"] = {}
    for key in ("quantized", "full", "safetensors"):
        if key in files:
            try:
                desc = describe_weights(files[key], cfg_kw)
            except Exception as ex:                                      # the reader met a layout it does not know: THE finding of a first contact
                print("[reader] %s: FAILED to parse: %s" % (key, ex)); failed = True
                continue
            out["steps"]["weights"][key] = desc
            print("[reader] %s: %d parameters, finite %s, schemes (stored, granularity, symmetric, bits): %s" % (key, desc["n_params"], desc["finite"], desc["schemes"]))
            for name in ("embeddings.word_embeddings.weight", "encoder.layer.0.attention.self.query.weight", "encoder.layer.0.intermediate.dense.weight", "encoder.layer.1.output.dense.weight", "encoder.layer.5.output.dense.weight"):
                if name in desc["tensors"]:
                    print("         %-52s %s" % (name, {k: v for k, v in desc["tensors"][name].items() if k not in ("scale_min", "scale_max")}))
    try:
        import torch
        have_gpu = torch.cuda.is_available()
    except Exception:
        have_gpu = False
    if have_gpu and ("quantized" in files or "full" in files):
        g = gpu_checks(files, cfg_kw)
        out["steps"]["gpu"] = g
        for k, v in g.items():
            print("[gpu] %-44s %s" % (k, v))
        failed |= not g["ok"]
    else:
        print("[gpu] skipped (no GPU or no model file)")
    if os.environ.get("SHODH_VERIFY_JSON"):
        json.dump(out, open(os.environ["SHODH_VERIFY_JSON"], "w"), indent=1)
    return 1 if failed else 0


if __name__ == "__main__":
    sys.exit(main(sys.argv))
