"""numpy restatement of the DYNAMICALLY QUANTISED MiniLM graph (TEST INFRASTRUCTURE ONLY; like everything under oracle/).

The reference's default model is `model_quint8_avx2.onnx` (embeddings/downloader.rs:31; minilm.rs:212-220), the ONNX Runtime
dynamic-quantisation export of all-MiniLM-L6-v2. Its arithmetic lives in ONNX Runtime 1.23.2, which is not in this image, and
the checkpoint cannot be downloaded: **parity with that file is UNPINNED**. What is restated here is the published semantics of
the ONNX operators that export is made of (ONNX operator spec + onnxruntime quantization tool defaults), applied to the BERT
graph of SURVEY.md Appendix E:

  DynamicQuantizeLinear (opset 11), per tensor, uint8:
      rmin = min(0, min(x)); rmax = max(0, max(x)); scale = (rmax - rmin) / 255   (1.0 if rmax == rmin)
      zp = round_half_even(clip(0 - rmin / scale, 0, 255));  q = clip(round_half_even(x / scale) + zp, 0, 255)
      -- over the WHOLE tensor [B, max_len, H], padding included (minilm.rs:588-593: "derives its activation scale from
         the whole tensor, padding included, so the tensor length is part of the embedding function")
  weights: per tensor, symmetric, 8 bit (quantize_dynamic, avx2 config): scale = 2 max|w| / 255, zp = 128 (uint8), i.e. the
      signed value clip(round_half_even(w / scale), -128, 127)
  MatMulInteger: int32 sum of (a - a_zp) * (b - b_zp), exact
  dequantise: float(acc) * (a_scale * b_scale) + bias            (MatMulIntegerToFloat)
  only MatMuls with a constant weight are quantised (MatMulConstBOnly, the dynamic-mode default): Q, K, V, attention output,
  FFN up, FFN down; QK^T, softmax, PV, GELU (erf), LayerNorm stay fp32. The word-embedding table is stored 8-bit (Gather) and
  dequantised on lookup.

Everything is float32 arithmetic; integer accumulators are exact. `tests/golden/make_int8_golden.py` runs this file to produce
`tests/golden/encoder_int8_golden.npz`.
"""
import math

import numpy as np

f32 = np.float32


def quantize_weight(w):
    """per-tensor symmetric 8-bit: -> (signed values int8, scale f32). The uint8 form is q + 128 with zero point 128."""
    w = np.asarray(w, f32)
    absmax = f32(np.max(np.abs(w))) if w.size else f32(0)
    scale = f32(2.0) * absmax / f32(255.0) if absmax > 0 else f32(1.0)
    q = np.clip(np.rint(w / scale), -128, 127).astype(np.int8)
    return q, f32(scale)


def dynamic_quantize_params(x):
    rmin = min(f32(0), f32(x.min())) if x.size else f32(0)
    rmax = max(f32(0), f32(x.max())) if x.size else f32(0)
    scale = f32(1.0) if rmax == rmin else f32(f32(rmax - rmin) / f32(255.0))
    zp = f32(0) - f32(rmin / scale)
    zp = int(np.rint(np.clip(zp, f32(0), f32(255))))
    return f32(scale), zp


def dynamic_quantize(x):
    """DynamicQuantizeLinear: -> (uint8 tensor, scale, zero point)"""
    x = np.asarray(x, f32)
    scale, zp = dynamic_quantize_params(x)
    q = np.clip(np.rint(x / scale) + f32(zp), 0, 255).astype(np.uint8)
    return q, scale, zp


def matmul_integer(a_u8, a_zp, w_s8):
    """exact int32 accumulators of sum_k (a - a_zp) * w   (a [M,K] uint8, w [N,K] signed)"""
    # float64 BLAS is exact here (|acc| < 2^53) and far faster than numpy's integer matmul
    acc = (a_u8.astype(np.float64) - float(a_zp)) @ w_s8.astype(np.float64).T
    return acc.astype(np.int64).astype(np.int32)


def dense_int8(x, w_q, w_scale, bias):
    """one quantised dense layer; x [..., K] f32. -> (y f32, acc int32, a_scale, a_zp)"""
    shp = x.shape
    a, sa, zp = dynamic_quantize(x)
    acc = matmul_integer(a.reshape(-1, shp[-1]), zp, w_q)
    y = acc.astype(f32) * f32(sa * w_scale) + bias.astype(f32)
    return y.reshape(shp[:-1] + (w_q.shape[0],)), acc, sa, zp


def layer_norm(x, g, b, eps):
    x = x.astype(f32)
    mean = x.mean(-1, keepdims=True, dtype=f32)
    d = x - mean
    var = (d * d).mean(-1, keepdims=True, dtype=f32)
    return (d / np.sqrt(var + f32(eps))) * g + b


_erf = np.vectorize(math.erf, otypes=[np.float64])


def gelu_erf(x):
    return (f32(0.5) * x * (f32(1.0) + _erf(x.astype(np.float64) * 0.7071067811865476).astype(f32))).astype(f32)


def quantize_model(sd, layers):
    """-> dict of quantised tensors (name -> (int8 values, scale)) for the word table and the six dense weights per layer"""
    q = {"embeddings.word_embeddings.weight": quantize_weight(sd["embeddings.word_embeddings.weight"])}
    for l in range(layers):
        p = "encoder.layer.%d." % l
        for nm in ("attention.self.query", "attention.self.key", "attention.self.value", "attention.output.dense", "intermediate.dense", "output.dense"):
            q[p + nm + ".weight"] = quantize_weight(sd[p + nm + ".weight"])
    return q


def encode(sd, ids, mask, heads=12, eps=1e-12, layers=6, trace=None):
    """ids/mask [B, S] (S = the padded tensor length, max_len). Rows with an empty mask are left out of the tensor (the
    reference never runs empty texts, minilm.rs:1123-1125, :1319-1350) and come back as zeros. -> unit vectors [B, H].
    `trace` (dict) receives the int32 accumulators and quantisation parameters of layer 0's query projection."""
    ids = np.asarray(ids); mask = np.asarray(mask)
    B, S = ids.shape
    H = sd["embeddings.word_embeddings.weight"].shape[1]
    out = np.zeros((B, H), f32)
    keep = np.nonzero(mask.sum(1) > 0)[0]
    if len(keep) == 0:
        return out
    ids, mask = ids[keep], mask[keep]
    Bk = len(keep)
    qm = quantize_model(sd, layers)
    wq, ws = qm["embeddings.word_embeddings.weight"]
    word = wq[np.clip(ids, 0, wq.shape[0] - 1)].astype(f32) * ws                       # Gather + DequantizeLinear
    x = word + sd["embeddings.position_embeddings.weight"][None, :S].astype(f32) + sd["embeddings.token_type_embeddings.weight"][0][None, None].astype(f32)
    x = layer_norm(x, sd["embeddings.LayerNorm.weight"], sd["embeddings.LayerNorm.bias"], eps).astype(f32)
    dh = H // heads
    lens = mask.sum(1)
    keymask = (np.arange(S)[None, :] < lens[:, None])
    for l in range(layers):
        p = "encoder.layer.%d." % l

        def dense(name, t):
            w_q, w_s = qm[p + name + ".weight"]
            return dense_int8(t, w_q, w_s, sd[p + name + ".bias"])
        q, acc_q, sa, zp = dense("attention.self.query", x)
        if trace is not None and l == 0:
            trace.update(acc_q=acc_q, a_scale=sa, a_zp=zp, w_scale=qm[p + "attention.self.query.weight"][1])
        k = dense("attention.self.key", x)[0]
        v = dense("attention.self.value", x)[0]
        qh = q.reshape(Bk, S, heads, dh).transpose(0, 2, 1, 3)
        kh = k.reshape(Bk, S, heads, dh).transpose(0, 2, 1, 3)
        vh = v.reshape(Bk, S, heads, dh).transpose(0, 2, 1, 3)
        sc = (qh @ kh.transpose(0, 1, 3, 2)).astype(f32) * f32(1.0 / math.sqrt(dh))
        sc = np.where(keymask[:, None, None, :], sc, f32(-3.0e38))                    # additive finfo.min mask == restriction to real keys
        sc = sc - sc.max(-1, keepdims=True)
        pr = np.exp(sc, dtype=f32)
        pr = pr / pr.sum(-1, keepdims=True, dtype=f32)
        ctx = (pr @ vh).astype(f32).transpose(0, 2, 1, 3).reshape(Bk, S, H)
        x = layer_norm(dense("attention.output.dense", ctx)[0] + x, sd[p + "attention.output.LayerNorm.weight"], sd[p + "attention.output.LayerNorm.bias"], eps).astype(f32)
        h = gelu_erf(dense("intermediate.dense", x)[0])
        x = layer_norm(dense("output.dense", h)[0] + x, sd[p + "output.LayerNorm.weight"], sd[p + "output.LayerNorm.bias"], eps).astype(f32)
    # masked mean-pool + finalize_pooled (minilm.rs:959-981, :846-878)
    m = keymask[:, :, None].astype(f32)
    pooled = (x * m).sum(1, dtype=f32) / lens[:, None].astype(f32)
    pooled = np.where(np.isfinite(pooled), pooled, f32(0))
    norm = np.sqrt((pooled * pooled).sum(1, keepdims=True, dtype=f32))
    pooled = np.where(norm > np.finfo(f32).eps, pooled / np.maximum(norm, f32(1e-30)), pooled)
    out[keep] = pooled.astype(f32)
    return out
