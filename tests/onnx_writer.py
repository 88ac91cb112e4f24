"""A minimal ONNX (protobuf wire format) WRITER for tests: produces files shaped like the two exports the reference loads
(embeddings/downloader.rs:29-53: onnx/model.onnx and onnx/model_quint8_avx2.onnx of all-MiniLM-L6-v2), from a state dict.

The `onnx` package is not in the image and the real files cannot be fetched, so the reader in csrc/weights_io.hip is exercised on
files written here, following what torch.onnx.export + onnxruntime.quantization.quantize_dynamic emit:
  * Linear layers become MatMul(x, W^T) + Add(bias, .): the weight is an ANONYMOUS transposed constant ("onnx::MatMul_<n>", [K, N]),
    the bias keeps its HF name;
  * quantize_dynamic replaces MatMul by DynamicQuantizeLinear -> MatMulInteger -> Cast -> Mul(a_scale * w_scale) -> Add(bias) and
    stores <w>_quantized (uint8 / int8), <w>_scale, <w>_zero_point (scalars, or [N] with per_channel=True); the word table's Gather
    reads a quantised table followed by DequantizeLinear.
Only what the reader looks at is written (graph.node with op_type / inputs / outputs / name, graph.initializer); value_info, opset
imports and attributes are left out. Field numbers: ModelProto.ir_version 1, .graph 7; GraphProto.node 1, .name 2, .initializer 5;
NodeProto.input 1, .output 2, .name 3, .op_type 4; TensorProto.dims 1, .data_type 2, .float_data 4, .int32_data 5, .name 8, .raw_data 9.
"""
import struct

import numpy as np

FLOAT, UINT8, INT8, FLOAT16 = 1, 2, 3, 10


def _varint(v):
    v &= (1 << 64) - 1
    out = bytearray()
    while True:
        b = v & 0x7F
        v >>= 7
        if v:
            out.append(b | 0x80)
        else:
            out.append(b)
            return bytes(out)


def _key(field, wt):
    return _varint((field << 3) | wt)


def _ld(field, payload):
    return _key(field, 2) + _varint(len(payload)) + payload


def _vi(field, v):
    return _key(field, 0) + _varint(v)


def tensor(name, arr, raw=True, packed_dims=False):
    arr = np.asarray(arr)
    dt = {np.dtype(np.float32): FLOAT, np.dtype(np.uint8): UINT8, np.dtype(np.int8): INT8, np.dtype(np.float16): FLOAT16}[arr.dtype]
    out = b""
    if packed_dims:
        out += _ld(1, b"".join(_varint(d) for d in arr.shape))
    else:
        for d in arr.shape:
            out += _vi(1, d)
    out += _vi(2, dt)
    if raw:
        out += _ld(9, np.ascontiguousarray(arr).tobytes())
    elif dt == FLOAT:
        out += _ld(4, np.ascontiguousarray(arr, np.float32).tobytes())             # packed float_data
    else:
        vals = arr.reshape(-1).view(np.uint16) if dt == FLOAT16 else arr.reshape(-1)
        out += _ld(5, b"".join(_varint(int(v)) for v in vals))                      # packed int32_data (negative int8: 10-byte varints)
    out += _ld(8, name.encode())
    return out


def node(op, inputs, outputs, name=""):
    out = b"".join(_ld(1, i.encode()) for i in inputs) + b"".join(_ld(2, o.encode()) for o in outputs)
    if name:
        out += _ld(3, name.encode())
    return out + _ld(4, op.encode())


def model(nodes, initializers):
    g = b"".join(_ld(1, n) for n in nodes) + _ld(2, b"main_graph") + b"".join(_ld(5, t) for t in initializers)
    return _vi(1, 8) + _ld(2, b"tests/onnx_writer.py") + _ld(7, g)


DENSE = ("attention.self.query", "attention.self.key", "attention.self.value", "attention.output.dense", "intermediate.dense", "output.dense")


def write_bert(path, sd, layers, qmodel=None, raw=True, named_bias=True, prefix="", packed_dims=False):
    """sd: HF BertModel names -> f32 arrays. qmodel (optional): name -> (q uint8|int8 [N, K], scale [1|N], zp [1|N]) for the tensors stored
    quantised (as oracle.int8_ref.quantize_model returns them). named_bias=False writes anonymous biases too (role only by graph order)."""
    nodes, inits = [], []
    ctr = [100]

    def anon(kind):
        ctr[0] += 1
        return "onnx::%s_%d" % (kind, ctr[0])

    def emb(name):
        full = prefix + name
        if qmodel is not None and name in qmodel:
            q, sc, zp = qmodel[name]
            inits.append(tensor(full + "_quantized", q, raw, packed_dims))
            inits.append(tensor(full + "_scale", np.asarray(sc, np.float32).reshape(()) if np.size(sc) == 1 else np.asarray(sc, np.float32), raw))
            inits.append(tensor(full + "_zero_point", np.asarray(zp, q.dtype).reshape(()) if np.size(zp) == 1 else np.asarray(zp, q.dtype), raw))
            nodes.append(node("Gather", [full + "_quantized", "input_ids"], [full + "_g"]))
            nodes.append(node("DequantizeLinear", [full + "_g", full + "_scale", full + "_zero_point"], [full + "_out"]))
        else:
            inits.append(tensor(full, sd[name].astype(np.float32), raw, packed_dims))
            nodes.append(node("Gather", [full, "input_ids"], [full + "_out"]))
    emb("embeddings.word_embeddings.weight")
    emb("embeddings.position_embeddings.weight")
    emb("embeddings.token_type_embeddings.weight")
    for nm in ("embeddings.LayerNorm.weight", "embeddings.LayerNorm.bias"):
        inits.append(tensor(prefix + nm, sd[nm].astype(np.float32), raw))
    x = "emb_ln_out"
    for l in range(layers):
        p = "encoder.layer.%d." % l
        for d in DENSE:
            wname, bname = p + d + ".weight", p + d + ".bias"
            b_init = prefix + bname if named_bias else anon("Add")
            inits.append(tensor(b_init, sd[bname].astype(np.float32), raw))
            out = "%s_out" % (p + d)
            if qmodel is not None and wname in qmodel:
                q, sc, zp = qmodel[wname]
                w = anon("MatMul")
                inits.append(tensor(w + "_quantized", np.ascontiguousarray(q.T), raw, packed_dims))          # ONNX MatMul B: [K, N]
                inits.append(tensor(w + "_scale", np.asarray(sc, np.float32).reshape(()) if np.size(sc) == 1 else np.asarray(sc, np.float32), raw))
                inits.append(tensor(w + "_zero_point", np.asarray(zp, q.dtype).reshape(()) if np.size(zp) == 1 else np.asarray(zp, q.dtype), raw))
                nodes.append(node("DynamicQuantizeLinear", [x], [x + "_q", x + "_s", x + "_z"]))
                nodes.append(node("MatMulInteger", [x + "_q", w + "_quantized", x + "_z", w + "_zero_point"], [out + "_mi"]))
                nodes.append(node("Cast", [out + "_mi"], [out + "_cf"]))
                nodes.append(node("Mul", [x + "_s", w + "_scale"], [out + "_sm"]))
                nodes.append(node("Mul", [out + "_cf", out + "_sm"], [out + "_dq"]))
                nodes.append(node("Add", [b_init, out + "_dq"], [out]))
            else:
                w = anon("MatMul")
                inits.append(tensor(w, np.ascontiguousarray(sd[wname].astype(np.float32).T), raw, packed_dims))
                nodes.append(node("MatMul", [x, w], [out + "_mm"]))
                nodes.append(node("Add", [b_init, out + "_mm"], [out]))
        for nm in ("attention.output.LayerNorm.weight", "attention.output.LayerNorm.bias", "output.LayerNorm.weight", "output.LayerNorm.bias"):
            inits.append(tensor(prefix + p + nm, sd[p + nm].astype(np.float32), raw))
    with open(path, "wb") as f:
        f.write(model(nodes, inits))


def write_safetensors(path, sd, dtype="F32", prefix="", extra=None):
    """the safetensors container by hand (8-byte header length, JSON header, data), so that F16 / BF16 and prefixes can be produced"""
    import json
    header, blobs, off = {"__metadata__": {"format": "pt"}}, [], 0
    items = list(sd.items()) + list((extra or {}).items())
    for name, a in items:
        a = np.ascontiguousarray(a, np.float32)
        if dtype == "F32":
            b = a.tobytes()
        elif dtype == "F16":
            b = a.astype(np.float16).tobytes()
        else:                                    # BF16: round to nearest even on the upper 16 bits
            u = a.view(np.uint32).astype(np.uint64)
            u = ((u + 0x7FFF + ((u >> 16) & 1)) >> 16).astype(np.uint16)
            b = u.tobytes()
        header[prefix + name] = {"dtype": dtype, "shape": list(a.shape), "data_offsets": [off, off + len(b)]}
        blobs.append(b)
        off += len(b)
    h = json.dumps(header, separators=(",", ":")).encode()
    h += b" " * ((8 - len(h) % 8) % 8)
    with open(path, "wb") as f:
        f.write(struct.pack("<Q", len(h)) + h + b"".join(blobs))
