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
  weights: whatever the export holds -- uint8 or int8 bytes with a scale and a zero point per tensor or per output channel
      (`quantize_weight_ort` restates the onnxruntime quantiser's rule for producing them: symmetric or asymmetric, optional 7-bit
      "reduce_range"; WHICH of these model_quint8_avx2.onnx uses cannot be checked offline, so the restatement and the HIP path take the
      tensors as given). The default when only f32 weights exist: per tensor, symmetric, scale = 2 max|w| / 255, zero point 128 in
      uint8 terms, i.e. the signed value clip(round_half_even(w / scale), -128, 127) with zero point 0 (`quantize_weight`).
  MatMulInteger: int32 sum of (a - a_zp) * (b - b_zp), exact, b_zp a scalar or one value per output channel
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


def quantize_weight_ort(w, per_channel=False, symmetric=False, reduce_range=False, signed=False):
    """The onnxruntime quantiser's weight rule (onnxruntime/python/tools/quantization/quant_utils.py: compute_scale_zp + quantize_nparray),
    restated: the range is widened to contain 0 (and made symmetric if asked), scale = (rmax - rmin) / (qmax - qmin),
    zero_point = round_half_even(qmin - rmin / scale), q = clip(round_half_even(w / scale) + zero_point, qmin, qmax).
    w [N, K] (HF layout); per_channel = one (scale, zero point) per output feature n. -> (q uint8|int8 [N, K], scale f32 [1|N], zp same dtype [1|N])"""
    w = np.asarray(w, f32)
    if signed:
        qmin, qmax = (-64, 64) if reduce_range else (-128, 127)
        if symmetric:
            qmin = -qmax if not reduce_range else qmin
    else:
        qmin, qmax = (0, 127) if reduce_range else (0, 255)
    rows = w if per_channel else w.reshape(1, -1)
    rmin = np.minimum(rows.min(1), f32(0)).astype(f32); rmax = np.maximum(rows.max(1), f32(0)).astype(f32)
    if symmetric:
        am = np.maximum(np.abs(rmin), np.abs(rmax)).astype(f32)
        rmin, rmax = -am, am
    scale = ((rmax - rmin) / f32(qmax - qmin)).astype(f32)
    tiny = scale < np.finfo(f32).tiny
    scale = np.where(tiny, f32(1), scale).astype(f32)
    zp = np.where(tiny, 0, np.rint(f32(qmin) - rmin / scale)).astype(np.int64)
    sc_full = scale[:, None] if per_channel else scale[0]
    zp_full = zp[:, None] if per_channel else zp[0]
    q = np.clip(np.rint(w / sc_full) + zp_full, qmin, qmax)
    dt = np.int8 if signed else np.uint8
    return q.astype(dt), scale.astype(f32), zp.astype(dt)


def as_triple(q, scale, zp=None):
    """(bytes, scales, zero points) with array-shaped scale / zero point"""
    q = np.asarray(q)
    scale = np.atleast_1d(np.asarray(scale, f32))
    zp = np.zeros(scale.shape, q.dtype) if zp is None else np.atleast_1d(np.asarray(zp)).astype(q.dtype)
    return q, scale, zp


def dequantize(q, scale, zp):
    """DequantizeLinear: (q - zp) * scale, per tensor or per row"""
    q, scale, zp = as_triple(q, scale, zp)
    sc = scale[:, None] if scale.size > 1 else scale[0]
    z = zp.astype(np.int32)[:, None] if zp.size > 1 else int(zp[0])
    return ((q.astype(np.int32) - z).astype(f32) * sc).astype(f32)


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


def matmul_integer(a_u8, a_zp, w_q, w_zp=None):
    """MatMulInteger: exact int32 accumulators of sum_k (a - a_zp) * (w - w_zp)   (a [M,K] uint8, w [N,K] uint8 or int8, w_zp scalar or [N])"""
    # float64 BLAS is exact here (|acc| < 2^53) and far faster than numpy's integer matmul
    wz = w_q.astype(np.float64)
    if w_zp is not None:
        z = np.atleast_1d(np.asarray(w_zp)).astype(np.float64)
        wz = wz - (z[:, None] if z.size > 1 else z[0])
    acc = (a_u8.astype(np.float64) - float(a_zp)) @ wz.T
    return acc.astype(np.int64).astype(np.int32)


def dense_int8(x, w_q, w_scale, bias, w_zp=None):
    """one quantised dense layer; x [..., K] f32; w_scale / w_zp scalars or [N]. -> (y f32, acc int32, a_scale, a_zp)"""
    shp = x.shape
    a, sa, zp = dynamic_quantize(x)
    acc = matmul_integer(a.reshape(-1, shp[-1]), zp, w_q, w_zp)
    ws = np.atleast_1d(np.asarray(w_scale, f32))
    y = acc.astype(f32) * (f32(sa) * (ws[None, :] if ws.size > 1 else ws[0])).astype(f32) + bias.astype(f32)
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


QUANTISED_NAMES = ("attention.self.query", "attention.self.key", "attention.self.value", "attention.output.dense", "intermediate.dense", "output.dense")


def quantize_model(sd, layers, rule=quantize_weight, word_rule=None):
    """-> dict name -> (bytes, scales, zero points) for the word table and the six dense weights per layer. `rule` maps an f32 matrix to
    (q, scale[, zp]): `quantize_weight` (this library's fallback for f32 checkpoints) or e.g. lambda w: quantize_weight_ort(w, per_channel=True)"""
    word_rule = word_rule or rule
    q = {"embeddings.word_embeddings.weight": as_triple(*word_rule(sd["embeddings.word_embeddings.weight"]))}
    for l in range(layers):
        p = "encoder.layer.%d." % l
        for nm in QUANTISED_NAMES:
            q[p + nm + ".weight"] = as_triple(*rule(sd[p + nm + ".weight"]))
    return q


def encode(sd, ids, mask, heads=12, eps=1e-12, layers=6, trace=None, qmodel=None, per_text=False):
    """ids/mask [B, S] (S = the padded tensor length, max_len). Rows with an empty mask are left out of the tensor (the
    reference never runs empty texts, minilm.rs:1123-1125, :1319-1350) and come back as zeros. -> unit vectors [B, H].
    `trace` (dict) receives the int32 accumulators and quantisation parameters of layer 0's query projection.
    per_text=False: ONE tensor [B, S, H] per DynamicQuantizeLinear = the reference's encode_batch (minilm.rs:996-1115).
    per_text=True: B tensors [1, S, H] = B calls of the reference's encode() (minilm.rs:883-982; what remember / recall run,
    memory/mod.rs:1037, retrieval.rs:673, :708, :878) -- restated literally as B calls of this function on one row each."""
    ids = np.asarray(ids); mask = np.asarray(mask)
    B, S = ids.shape
    if per_text and B > 1:
        qm = qmodel if qmodel is not None else quantize_model(sd, layers)
        return np.concatenate([encode(sd, ids[i:i + 1], mask[i:i + 1], heads, eps, layers, None, qm) for i in range(B)], 0)
    H = sd["embeddings.word_embeddings.weight"].shape[1]
    out = np.zeros((B, H), f32)
    keep = np.nonzero(mask.sum(1) > 0)[0]
    if len(keep) == 0:
        return out
    ids, mask = ids[keep], mask[keep]
    Bk = len(keep)
    qm = qmodel if qmodel is not None else quantize_model(sd, layers)                   # an export's tensors, or the f32 weights quantised by the fallback rule
    wq, ws, wz = qm["embeddings.word_embeddings.weight"]
    word = (wq[np.clip(ids, 0, wq.shape[0] - 1)].astype(np.int32) - int(wz[0])).astype(f32) * ws[0]     # Gather + DequantizeLinear
    x = word + sd["embeddings.position_embeddings.weight"][None, :S].astype(f32) + sd["embeddings.token_type_embeddings.weight"][0][None, None].astype(f32)
    x = layer_norm(x, sd["embeddings.LayerNorm.weight"], sd["embeddings.LayerNorm.bias"], eps).astype(f32)
    dh = H // heads
    lens = mask.sum(1)
    keymask = (np.arange(S)[None, :] < lens[:, None])
    for l in range(layers):
        p = "encoder.layer.%d." % l

        def dense(name, t):
            w_q, w_s, w_z = qm[p + name + ".weight"]
            return dense_int8(t, w_q, w_s, sd[p + name + ".bias"], w_z)
        q, acc_q, sa, zp = dense("attention.self.query", x)
        if trace is not None and l == 0:
            trace.update(acc_q=acc_q, a_scale=sa, a_zp=zp, w_scale=qm[p + "attention.self.query.weight"][1][0])
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
