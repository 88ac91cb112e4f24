"""encoder-only timing: python tools/enc_bench.py [bf16|int8] [batch]  (prints one JSON line; SHODH_ENC_UNFUSED=1 for the round-1 FFN)"""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import torch
import bench
import shodh_memory_amd as S
from shodh_memory_amd import _lib as L
dev = torch.device("cuda", 0)
dname = sys.argv[1] if len(sys.argv) > 1 else "bf16"
b = int(sys.argv[2]) if len(sys.argv) > 2 else 4096
dtype = {"bf16": L.DTYPE_BF16, "int8": L.DTYPE_INT8, "fp32": L.DTYPE_FP32}[dname]
g = torch.Generator(device=dev).manual_seed(5)
export = os.environ.get("SHODH_ENC_EXPORT", "")      # "u8" / "u8pc": INT8 mode on a file written like a dynamic-quantisation export (uint8 weights with zero points, per tensor / per channel)
if dname == "int8" and export:
    import tempfile
    import numpy as np
    from oracle import int8_ref as R                   # (a tool, not the product: the restatement's quantiser writes the file)
    from shodh_memory_amd import embedder as E
    from tests import onnx_writer as W
    cfg = E.embed_cfg()
    sd = E.blob_to_state_dict(E.synthetic_weights(1234, cfg), cfg)
    for k in sd:
        if k.endswith("dense.weight") or k.endswith("query.weight") or k.endswith("key.weight") or k.endswith("value.weight"):
            sd[k] += np.float32(0.004)                 # asymmetric ranges: non-zero zero points
    qm = R.quantize_model(sd, cfg.layers, rule=(lambda w: R.quantize_weight_ort(w, per_channel=(export == "u8pc"))), word_rule=lambda w: R.quantize_weight_ort(w))
    path = os.path.join(tempfile.mkdtemp(), "model_quantized.onnx")
    W.write_bert(path, sd, cfg.layers, qmodel=qm)
    enc = S.MiniLMEmbedder(dtype=dtype, weights_path=path)
else:
    enc = S.MiniLMEmbedder(synthetic_seed=1234, dtype=dtype)
if os.environ.get("SHODH_ENC_PER_TEXT", "") == "1":   # SHODH_QUANT_SCOPE_PER_TEXT: one range per text
    enc.set_quant_scope(L.QUANT_SCOPE_PER_TEXT)
ids, mask, lens = bench.synth_tokens(torch, b, 256, g, dev)
emb = torch.empty((b, 384), dtype=torch.float32, device=dev)
dt = bench.timed_steps(torch, lambda i: enc.encode_ids_device(ids, mask, out=emb), 10, 3)
tokens = int(lens.sum()); H, F, LAY = 384, 1536, 6
tok_c = b * 256 if dname == "int8" else tokens
att = float((lens.double() * 256).sum()) if dname == "int8" else float((lens.double() ** 2).sum())
flop = float(tok_c * 2 * (4 * H * H + 2 * H * F) * LAY + att * 4 * H * LAY)
print(json.dumps({"encoder": dname, "batch": b, "ms": round(dt * 1e3, 3), "texts_per_s": round(b / dt, 1), "tokens": tokens, "tflops": round(flop / dt / 1e12, 1),
                  "unfused": os.environ.get("SHODH_ENC_UNFUSED", "0"), "weights": export or "synthetic",
                  "quant_scope": "per_text" if os.environ.get("SHODH_ENC_PER_TEXT", "") == "1" else "batch"}))
