// weights_io.hip -- readers of the encoder's weight files (host code only; no device call in this file).
//
// What the reference loads (minilm.rs:212-220; downloader.rs:29-53): `model_quantized.onnx` = onnx/model_quint8_avx2.onnx of
// sentence-transformers/all-MiniLM-L6-v2 (the ONNX Runtime dynamic-quantisation export, the DEFAULT) or `model.onnx` (fp32). The
// checkpoint itself ships as model.safetensors. This file reads all three containers without ONNX Runtime / protobuf / serde:
//   * safetensors: 8-byte little-endian header length, a JSON header {name: {dtype, shape, data_offsets}}, raw little-endian data;
//   * ONNX: the protobuf wire format of ModelProto.graph (field 7) -> GraphProto.node (1) / .initializer (5) -> TensorProto
//     (dims 1, data_type 2, float_data 4, int32_data 5, name 8, raw_data 9). Parameters are found by their HF names where the
//     exporter kept them (embedding tables, LayerNorm, biases) and, for MatMul weights -- which torch.onnx.export stores as anonymous
//     transposed constants ("onnx::MatMul_1234", [K][N]) -- by walking MatMul / MatMulInteger -> (Cast, Mul)* -> Add(bias) and
//     reading the role off the bias' name; if the biases are anonymous too, by graph order (query, key, value, attention output,
//     intermediate, output per layer). A quantised tensor is the triple the onnxruntime quantiser writes: <w>_quantized (uint8 or
//     int8), <w>_scale, <w>_zero_point, per tensor or per output channel.
// Parity note: no real export is available offline (no network), so the reader is tested on files a committed script writes in
// these formats (tests/onnx_writer.py, safetensors 0.x from the image) -- see DESIGN.md section 2.
#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <map>
#include <new>
#include <stdexcept>
#include <string>
#include <vector>

#include "common.h"
#include "weights_io.h"

namespace shodh {

std::vector<TensorSlot> tensor_table(const shodh_embed_cfg &cfg, uint64_t *n_params) {
    std::vector<TensorSlot> t;
    size_t o = 0;
    const uint32_t H = cfg.hidden, I = cfg.intermediate;
    auto add = [&](const std::string &name, uint32_t rows, uint32_t cols, bool dense, bool quantisable) {
        TensorSlot s; s.name = name; s.offset = o; s.rows = rows; s.cols = cols; s.dense = dense; s.quantisable = quantisable;
        o += (size_t)rows * cols; t.push_back(s);
    };
    add("embeddings.word_embeddings.weight", cfg.vocab, H, false, true);
    add("embeddings.position_embeddings.weight", cfg.max_pos, H, false, false);
    add("embeddings.token_type_embeddings.weight", cfg.type_vocab, H, false, false);
    add("embeddings.LayerNorm.weight", 1, H, false, false);
    add("embeddings.LayerNorm.bias", 1, H, false, false);
    for (uint32_t l = 0; l < cfg.layers; ++l) {
        const std::string p = "encoder.layer." + std::to_string(l) + ".";
        for (const char *nm : {"query", "key", "value"}) {
            add(p + "attention.self." + nm + ".weight", H, H, true, true);
            add(p + "attention.self." + nm + ".bias", 1, H, false, false);
        }
        add(p + "attention.output.dense.weight", H, H, true, true); add(p + "attention.output.dense.bias", 1, H, false, false);
        add(p + "attention.output.LayerNorm.weight", 1, H, false, false); add(p + "attention.output.LayerNorm.bias", 1, H, false, false);
        add(p + "intermediate.dense.weight", I, H, true, true); add(p + "intermediate.dense.bias", 1, I, false, false);
        add(p + "output.dense.weight", H, I, true, true); add(p + "output.dense.bias", 1, H, false, false);
        add(p + "output.LayerNorm.weight", 1, H, false, false); add(p + "output.LayerNorm.bias", 1, H, false, false);
    }
    if (n_params) *n_params = o;
    return t;
}

void WeightSet::init(const shodh_embed_cfg &c) {
    cfg = c;
    uint64_t n = 0;
    slots = tensor_table(c, &n);
    blob.assign(n, 0.0f);
    have.assign(slots.size(), 0);
    q.assign(slots.size(), QTensor());
}
int WeightSet::find(const char *name) const {
    if (!name) return -1;
    for (size_t i = 0; i < slots.size(); ++i) if (slots[i].name == name) return (int)i;
    return -1;
}
int WeightSet::set_f32(int slot, const float *data, uint64_t n, bool transposed) {
    if (slot < 0 || slot >= (int)slots.size() || !data) { set_error("bad tensor slot"); return SHODH_ERR_INVALID; }
    const TensorSlot &s = slots[slot];
    if (n != (uint64_t)s.rows * s.cols) { set_error("%s: %llu values, expected %u x %u", s.name.c_str(), (unsigned long long)n, s.rows, s.cols); return SHODH_ERR_INVALID; }
    float *dst = blob.data() + s.offset;
    if (!transposed) memcpy(dst, data, n * 4);
    else for (uint32_t r = 0; r < s.rows; ++r) for (uint32_t c = 0; c < s.cols; ++c) dst[(size_t)r * s.cols + c] = data[(size_t)c * s.rows + r];
    have[slot] = 1; q[slot] = QTensor();
    return SHODH_OK;
}
int WeightSet::set_quantized(int slot, const void *data, bool is_signed, bool transposed, const float *scale, const void *zero_point, uint32_t n_scale) {
    if (slot < 0 || slot >= (int)slots.size() || !data || !scale) { set_error("bad tensor slot / null argument"); return SHODH_ERR_INVALID; }
    const TensorSlot &s = slots[slot];
    if (!s.quantisable) { set_error("%s is not a tensor a dynamic-quantisation export stores in 8 bits", s.name.c_str()); return SHODH_ERR_UNSUPPORTED; }
    if (n_scale != 1 && n_scale != s.rows) { set_error("%s: %u scales; expected 1 (per tensor) or %u (per output channel)", s.name.c_str(), n_scale, s.rows); return SHODH_ERR_INVALID; }
    if (!s.dense && n_scale != 1) { set_error("%s: the embedding table takes one scale", s.name.c_str()); return SHODH_ERR_UNSUPPORTED; }
    QTensor t; t.present = true; t.N = s.rows; t.K = s.cols; t.n_scale = n_scale;
    t.q.resize((size_t)s.rows * s.cols); t.scale.assign(scale, scale + n_scale); t.zp.resize(n_scale);
    for (uint32_t c = 0; c < n_scale; ++c) {
        if (!(t.scale[c] == t.scale[c]) || std::isinf(t.scale[c])) { set_error("%s: non-finite scale", s.name.c_str()); return SHODH_ERR_NONFINITE; }
        const int z = zero_point ? (is_signed ? (int)((const int8_t *)zero_point)[c] : (int)((const uint8_t *)zero_point)[c]) : 0;
        t.zp[c] = is_signed ? z : z - 128;
    }
    const uint8_t *u = (const uint8_t *)data;
    float *dst = blob.data() + s.offset;
    for (uint32_t n = 0; n < s.rows; ++n) {
        const uint32_t c = n_scale == 1 ? 0 : n;
        for (uint32_t k = 0; k < s.cols; ++k) {
            const uint8_t raw = transposed ? u[(size_t)k * s.rows + n] : u[(size_t)n * s.cols + k];
            const int v = is_signed ? (int)(int8_t)raw : (int)raw - 128;
            t.q[(size_t)n * s.cols + k] = (int8_t)v;
            dst[(size_t)n * s.cols + k] = (float)(v - t.zp[c]) * t.scale[c];      // DequantizeLinear: (x - zp) * scale
        }
    }
    q[slot] = std::move(t);
    have[slot] = 2;
    return SHODH_OK;
}
int WeightSet::check_complete() const {
    for (size_t i = 0; i < slots.size(); ++i)
        if (!have[i]) { set_error("weights incomplete: %s was not supplied", slots[i].name.c_str()); return SHODH_ERR_STATE; }
    return SHODH_OK;
}

namespace {

int slurp(const char *path, std::vector<unsigned char> &buf) {
    FILE *f = fopen(path, "rb");
    if (!f) { set_error("cannot open %s", path); return SHODH_ERR_IO; }
    fseek(f, 0, SEEK_END);
    const long n = ftell(f);
    fseek(f, 0, SEEK_SET);
    if (n < 0) { fclose(f); set_error("cannot size %s", path); return SHODH_ERR_IO; }
    buf.resize((size_t)n);
    const size_t got = n ? fread(buf.data(), 1, (size_t)n, f) : 0;
    fclose(f);
    if (got != (size_t)n) { set_error("short read on %s", path); return SHODH_ERR_IO; }
    return SHODH_OK;
}

float half_to_float(uint16_t h) {
    const uint32_t sign = (uint32_t)(h & 0x8000u) << 16, exp = (h >> 10) & 0x1F, man = h & 0x3FFu;
    uint32_t b;
    if (exp == 0) {
        if (man == 0) b = sign;
        else { int e = -1; uint32_t m = man; do { ++e; m <<= 1; } while (!(m & 0x400u)); b = sign | ((uint32_t)(127 - 15 - e) << 23) | ((m & 0x3FFu) << 13); }
    } else if (exp == 31) b = sign | 0x7F800000u | (man << 13);
    else b = sign | ((exp + 112) << 23) | (man << 13);
    float f; memcpy(&f, &b, 4); return f;
}
float bf16_to_float(uint16_t h) { const uint32_t b = (uint32_t)h << 16; float f; memcpy(&f, &b, 4); return f; }

// does `key` name the slot `want`, possibly behind a module prefix ("bert.", "0.auto_model.")?
bool name_matches(const std::string &key, const std::string &want) {
    if (key.size() < want.size()) return false;
    if (key.compare(key.size() - want.size(), want.size(), want) != 0) return false;
    return key.size() == want.size() || key[key.size() - want.size() - 1] == '.';
}

// ---- safetensors ----------------------------------------------------------------------------------------------------------
// a minimal JSON reader for the header: objects, arrays, strings (with escapes), numbers, true/false/null
struct Json {
    const char *p, *e;
    bool ok = true;
    void ws() { while (p < e && (*p == ' ' || *p == '\n' || *p == '\t' || *p == '\r')) ++p; }
    bool eat(char c) { ws(); if (p < e && *p == c) { ++p; return true; } return false; }
    std::string str() {
        std::string s; ws();
        if (p >= e || *p != '"') { ok = false; return s; }
        ++p;
        while (p < e && *p != '"') {
            if (*p == '\\' && p + 1 < e) {
                ++p;
                switch (*p) { case 'n': s += '\n'; break; case 't': s += '\t'; break; case 'r': s += '\r'; break; case 'b': s += '\b'; break; case 'f': s += '\f'; break;
                              case 'u': if (p + 4 < e) p += 4; s += '?'; break; default: s += *p; }
                ++p;
            } else s += *p++;
        }
        if (p >= e) { ok = false; return s; }
        ++p;
        return s;
    }
    // the numbers of a safetensors header are shapes and byte offsets: non-negative integers. Read digit by digit inside [p, e) -- the header is
    // not NUL-terminated, strtod would run past its end on a crafted file -- and anything else (sign, fraction, exponent, overflow) is malformed.
    uint64_t num() {
        ws();
        if (p >= e || *p < '0' || *p > '9') { ok = false; return 0; }
        uint64_t v = 0;
        while (p < e && *p >= '0' && *p <= '9') {
            const uint64_t d = (uint64_t)(*p - '0');
            if (v > (UINT64_MAX - d) / 10) { ok = false; return 0; }
            v = v * 10 + d; ++p;
        }
        if (p < e && (*p == '.' || *p == 'e' || *p == 'E')) { ok = false; return 0; }
        return v;
    }
    void skip(int depth = 0) {          // any value (nesting bounded: a crafted __metadata__ must not overflow the stack)
        ws();
        if (p >= e || depth > 32) { ok = false; return; }
        if (*p == '"') { str(); return; }
        if (*p == '{') { ++p; if (eat('}')) return; do { str(); if (!eat(':')) { ok = false; return; } skip(depth + 1); } while (ok && eat(',')); if (!eat('}')) ok = false; return; }
        if (*p == '[') { ++p; if (eat(']')) return; do { skip(depth + 1); } while (ok && eat(',')); if (!eat(']')) ok = false; return; }
        while (p < e && *p != ',' && *p != '}' && *p != ']' && *p != ' ' && *p != '\n') ++p;
    }
};
struct StEntry { std::string dtype; std::vector<uint64_t> shape; uint64_t b = 0, e = 0; };

int load_safetensors(const std::vector<unsigned char> &buf, const char *path, WeightSet &ws) {
    if (buf.size() < 8) { set_error("%s: not a safetensors file (shorter than its length field)", path); return SHODH_ERR_IO; }
    uint64_t hl; memcpy(&hl, buf.data(), 8);
    if (hl > buf.size() - 8 || hl < 2) { set_error("%s: safetensors header length %llu exceeds the file", path, (unsigned long long)hl); return SHODH_ERR_IO; }
    Json j{(const char *)buf.data() + 8, (const char *)buf.data() + 8 + hl};
    const unsigned char *data = buf.data() + 8 + hl;
    const uint64_t data_len = buf.size() - 8 - hl;
    std::map<std::string, StEntry> ent;
    if (!j.eat('{')) { set_error("%s: safetensors header is not a JSON object", path); return SHODH_ERR_IO; }
    if (!j.eat('}')) {
        do {
            const std::string key = j.str();
            if (!j.ok || !j.eat(':')) { set_error("%s: malformed safetensors header", path); return SHODH_ERR_IO; }
            if (key == "__metadata__") { j.skip(); continue; }
            StEntry en;
            if (!j.eat('{')) { set_error("%s: malformed entry %s", path, key.c_str()); return SHODH_ERR_IO; }
            do {
                const std::string f = j.str();
                if (!j.eat(':')) { j.ok = false; break; }
                if (f == "dtype") en.dtype = j.str();
                else if (f == "shape") { if (!j.eat('[')) { j.ok = false; break; } if (!j.eat(']')) { do { en.shape.push_back(j.num()); } while (j.ok && j.eat(',')); if (!j.eat(']')) j.ok = false; } }
                else if (f == "data_offsets") { if (!j.eat('[')) { j.ok = false; break; } en.b = j.num(); if (!j.eat(',')) j.ok = false; en.e = j.num(); if (!j.eat(']')) j.ok = false; }
                else j.skip();
            } while (j.ok && j.eat(','));
            if (!j.ok || !j.eat('}')) { set_error("%s: malformed entry %s", path, key.c_str()); return SHODH_ERR_IO; }
            ent[key] = en;
        } while (j.ok && j.eat(','));
        if (!j.ok || !j.eat('}')) { set_error("%s: malformed safetensors header", path); return SHODH_ERR_IO; }
    }
    std::vector<float> tmp;
    for (size_t si = 0; si < ws.slots.size(); ++si) {
        const TensorSlot &s = ws.slots[si];
        const StEntry *en = nullptr;
        for (auto &kv : ent) if (name_matches(kv.first, s.name)) { en = &kv.second; break; }
        if (!en) continue;            // check_complete names it
        uint64_t n = 1;
        bool wrapped = false;
        for (uint64_t d : en->shape) { if (d && n > UINT64_MAX / d) wrapped = true; n *= d; }
        const uint64_t want = (uint64_t)s.rows * s.cols;
        if (wrapped) { set_error("%s: tensor %s has an impossible shape", path, s.name.c_str()); return SHODH_ERR_INVALID; }
        const size_t esz = en->dtype == "F32" ? 4 : (en->dtype == "F16" || en->dtype == "BF16") ? 2 : 0;
        if (!esz) { set_error("%s: tensor %s has dtype %s (F32 / F16 / BF16 supported)", path, s.name.c_str(), en->dtype.c_str()); return SHODH_ERR_UNSUPPORTED; }
        if (n != want || en->e < en->b || en->e > data_len || en->e - en->b != n * esz) {
            set_error("%s: tensor %s has %llu elements / %llu bytes, expected %u x %u", path, s.name.c_str(), (unsigned long long)n, (unsigned long long)(en->e - en->b), s.rows, s.cols);
            return SHODH_ERR_INVALID;
        }
        tmp.resize(n);
        const unsigned char *src = data + en->b;
        if (esz == 4) memcpy(tmp.data(), src, n * 4);
        else for (uint64_t i = 0; i < n; ++i) { uint16_t h; memcpy(&h, src + 2 * i, 2); tmp[i] = en->dtype == "F16" ? half_to_float(h) : bf16_to_float(h); }
        SHODH_TRY(ws.set_f32((int)si, tmp.data(), n, false));
    }
    return ws.check_complete();
}

// ---- ONNX (protobuf wire format) ----------------------------------------------------------------------------------------------
struct Pb {
    const unsigned char *p, *e;
    bool ok = true;
    bool more() const { return ok && p < e; }
    uint64_t varint() {
        uint64_t v = 0; int sh = 0;
        while (p < e && sh < 70) { const unsigned char c = *p++; v |= (uint64_t)(c & 0x7F) << sh; if (!(c & 0x80)) return v; sh += 7; }
        ok = false; return 0;
    }
    // reads a tag; for length-delimited fields [sp, se) is the payload; for varints `val`; fixed-width fields are skipped into val
    bool field(uint32_t &num, uint32_t &wt, uint64_t &val, const unsigned char *&sp, const unsigned char *&se) {
        if (!more()) return false;
        const uint64_t tag = varint();
        if (!ok) return false;
        num = (uint32_t)(tag >> 3); wt = (uint32_t)(tag & 7);
        switch (wt) {
            case 0: val = varint(); return ok;
            case 1: if (e - p < 8) { ok = false; return false; } memcpy(&val, p, 8); p += 8; return true;
            case 5: if (e - p < 4) { ok = false; return false; } { uint32_t v; memcpy(&v, p, 4); val = v; } p += 4; return true;
            case 2: { const uint64_t n = varint(); if (!ok || n > (uint64_t)(e - p)) { ok = false; return false; } sp = p; se = p + n; p += n; return true; }
            default: ok = false; return false;
        }
    }
};
struct OnnxTensor {
    std::string name;
    std::vector<int64_t> dims;
    int dtype = 0;                       // 1 FLOAT, 2 UINT8, 3 INT8, 6 INT32, 7 INT64, 10 FLOAT16, 16 BFLOAT16
    const unsigned char *raw = nullptr; size_t raw_n = 0;
    std::vector<float> fdata;            // float_data (field 4)
    std::vector<int32_t> idata;          // int32_data (field 5): holds uint8 / int8 / float16 elements when raw_data is absent
    bool external = false;
    // UINT64_MAX = impossible (negative or overflowing dims): matches no payload size below
    uint64_t numel() const { uint64_t n = 1; for (int64_t d : dims) { if (d < 0) return UINT64_MAX; if (d && n > (UINT64_MAX - 1) / (uint64_t)d) return UINT64_MAX; n *= (uint64_t)d; } return n; }
};
struct OnnxNode { std::string op, name; std::vector<std::string> in, out; };

bool parse_tensor(const unsigned char *sp, const unsigned char *se, OnnxTensor &t) {
    Pb pb{sp, se};
    uint32_t num, wt; uint64_t val; const unsigned char *a, *b;
    while (pb.field(num, wt, val, a, b)) {
        switch (num) {
            case 1: if (wt == 0) t.dims.push_back((int64_t)val); else if (wt == 2) { Pb q{a, b}; while (q.more()) t.dims.push_back((int64_t)q.varint()); if (!q.ok) return false; } break;
            case 2: t.dtype = (int)val; break;
            case 4: if (wt == 2) { const size_t n = (size_t)(b - a) / 4; t.fdata.resize(n); memcpy(t.fdata.data(), a, n * 4); } else if (wt == 5) { float f; const uint32_t v = (uint32_t)val; memcpy(&f, &v, 4); t.fdata.push_back(f); } break;
            case 5: if (wt == 2) { Pb q{a, b}; while (q.more()) t.idata.push_back((int32_t)q.varint()); if (!q.ok) return false; } else if (wt == 0) t.idata.push_back((int32_t)val); break;
            case 8: if (wt == 2) t.name.assign((const char *)a, (size_t)(b - a)); break;
            case 9: if (wt == 2) { t.raw = a; t.raw_n = (size_t)(b - a); } break;
            case 13: t.external = true; break;
            case 14: if (val == 1) t.external = true; break;
            default: break;
        }
    }
    return pb.ok;
}
bool parse_node(const unsigned char *sp, const unsigned char *se, OnnxNode &n) {
    Pb pb{sp, se};
    uint32_t num, wt; uint64_t val; const unsigned char *a, *b;
    while (pb.field(num, wt, val, a, b)) {
        if (wt != 2) continue;
        if (num == 1) n.in.emplace_back((const char *)a, (size_t)(b - a));
        else if (num == 2) n.out.emplace_back((const char *)a, (size_t)(b - a));
        else if (num == 3) n.name.assign((const char *)a, (size_t)(b - a));
        else if (num == 4) n.op.assign((const char *)a, (size_t)(b - a));
    }
    return pb.ok;
}
// tensor -> f32 values (FLOAT / FLOAT16 / BFLOAT16)
bool tensor_floats(const OnnxTensor &t, std::vector<float> &out) {
    const uint64_t n = t.numel();
    // the element count comes from the file's dims: it is believed only when a payload of exactly that size is there (a crafted dim must not size an allocation)
    const size_t esz = t.dtype == 1 ? 4 : 2;
    const bool raw_ok = t.raw && n <= t.raw_n && (uint64_t)t.raw_n == n * esz;
    const bool typed_ok = t.dtype == 1 ? (uint64_t)t.fdata.size() == n : (uint64_t)t.idata.size() == n;
    if (!raw_ok && !typed_ok) return false;
    out.resize(n);
    if (t.dtype == 1) {
        if (t.raw && t.raw_n == n * 4) { memcpy(out.data(), t.raw, n * 4); return true; }
        if (t.fdata.size() == n) { out = t.fdata; return true; }
        return false;
    }
    if (t.dtype == 10 || t.dtype == 16) {
        for (uint64_t i = 0; i < n; ++i) {
            uint16_t h;
            if (t.raw && t.raw_n == n * 2) memcpy(&h, t.raw + 2 * i, 2); else if (t.idata.size() == n) h = (uint16_t)t.idata[i]; else return false;
            out[i] = t.dtype == 10 ? half_to_float(h) : bf16_to_float(h);
        }
        return true;
    }
    return false;
}
// tensor -> bytes (UINT8 / INT8)
bool tensor_bytes(const OnnxTensor &t, std::vector<uint8_t> &out) {
    const uint64_t n = t.numel();
    if (t.dtype != 2 && t.dtype != 3) return false;
    if (!(t.raw && (uint64_t)t.raw_n == n) && (uint64_t)t.idata.size() != n) return false;      // (before the allocation: see tensor_floats)
    out.resize(n);
    if (t.raw && t.raw_n == n) { memcpy(out.data(), t.raw, n); return true; }
    if (t.idata.size() == n) { for (uint64_t i = 0; i < n; ++i) out[i] = (uint8_t)t.idata[i]; return true; }
    return false;
}

struct OnnxGraph {
    std::vector<OnnxTensor> init;
    std::vector<OnnxNode> nodes;
    std::map<std::string, int> init_by_name;
    std::multimap<std::string, int> consumers;     // value name -> node index
    std::map<std::string, int> producer;           // value name -> node index
    const OnnxTensor *tensor(const std::string &n) const { auto it = init_by_name.find(n); return it == init_by_name.end() ? nullptr : &init[it->second]; }
};

std::string strip_suffix(const std::string &s, const char *suf) {
    const size_t n = strlen(suf);
    return (s.size() >= n && s.compare(s.size() - n, n, suf) == 0) ? s.substr(0, s.size() - n) : std::string();
}

// installs initialiser `base` (f32) or the triple base_quantized / base_scale / base_zero_point into slot si
int install_by_name(const OnnxGraph &g, const char *path, WeightSet &ws, int si, const std::string &base, bool transposed) {
    const TensorSlot &s = ws.slots[si];
    const uint64_t want = (uint64_t)s.rows * s.cols;
    if (const OnnxTensor *t = g.tensor(base)) {
        if (t->dtype == 1 || t->dtype == 10 || t->dtype == 16) {
            if (t->external) { set_error("%s: tensor %s keeps its data in an external file (unsupported)", path, base.c_str()); return SHODH_ERR_UNSUPPORTED; }
            std::vector<float> v;
            if (t->numel() != want || !tensor_floats(*t, v)) { set_error("%s: tensor %s does not hold %u x %u floats", path, base.c_str(), s.rows, s.cols); return SHODH_ERR_INVALID; }
            return ws.set_f32(si, v.data(), v.size(), transposed);
        }
    }
    const OnnxTensor *tq = g.tensor(base + "_quantized");
    if (!tq) tq = g.tensor(base);          // an initialiser that is already 8-bit under its own name
    if (tq && (tq->dtype == 2 || tq->dtype == 3)) {
        const std::string stem = strip_suffix(tq->name, "_quantized").empty() ? tq->name : strip_suffix(tq->name, "_quantized");
        const OnnxTensor *tsc = g.tensor(stem + "_scale"), *tzp = g.tensor(stem + "_zero_point");
        if (!tsc) { set_error("%s: %s has no %s_scale initialiser", path, tq->name.c_str(), stem.c_str()); return SHODH_ERR_INVALID; }
        std::vector<uint8_t> bytes, zp;
        std::vector<float> sc;
        if (tq->external || tq->numel() != want || !tensor_bytes(*tq, bytes) || !tensor_floats(*tsc, sc) || sc.empty()) { set_error("%s: quantised tensor %s is malformed", path, tq->name.c_str()); return SHODH_ERR_INVALID; }
        if (tzp) { if (tzp->dtype != tq->dtype || !tensor_bytes(*tzp, zp) || zp.size() != sc.size()) { set_error("%s: zero point of %s does not match its scale / type", path, tq->name.c_str()); return SHODH_ERR_INVALID; } }
        return ws.set_quantized(si, bytes.data(), tq->dtype == 3, transposed, sc.data(), tzp ? zp.data() : nullptr, (uint32_t)sc.size());
    }
    return 1;      // not found under this name (positive: not an error)
}

int load_onnx(const std::vector<unsigned char> &buf, const char *path, WeightSet &ws) {
    OnnxGraph g;
    {
        Pb model{buf.data(), buf.data() + buf.size()};
        uint32_t num, wt; uint64_t val; const unsigned char *a = nullptr, *b = nullptr;
        const unsigned char *ga = nullptr, *gb = nullptr;
        while (model.field(num, wt, val, a, b)) if (num == 7 && wt == 2) { ga = a; gb = b; }
        if (!model.ok || !ga) { set_error("%s: not an ONNX ModelProto (no graph field)", path); return SHODH_ERR_IO; }
        Pb gp{ga, gb};
        while (gp.field(num, wt, val, a, b)) {
            if (wt != 2) continue;
            if (num == 1) { OnnxNode n; if (!parse_node(a, b, n)) { gp.ok = false; break; } g.nodes.push_back(std::move(n)); }
            else if (num == 5) { OnnxTensor t; if (!parse_tensor(a, b, t)) { gp.ok = false; break; } g.init.push_back(std::move(t)); }
        }
        if (!gp.ok) { set_error("%s: malformed ONNX graph", path); return SHODH_ERR_IO; }
    }
    for (size_t i = 0; i < g.init.size(); ++i) g.init_by_name[g.init[i].name] = (int)i;
    for (size_t i = 0; i < g.nodes.size(); ++i) {
        for (auto &v : g.nodes[i].in) g.consumers.insert({v, (int)i});
        for (auto &v : g.nodes[i].out) g.producer[v] = (int)i;
    }
    // 1. everything the exporter kept under its HF name
    for (size_t si = 0; si < ws.slots.size(); ++si) {
        const std::string &want = ws.slots[si].name;
        for (auto &t : g.init) {
            std::string base = t.name;
            const std::string st = strip_suffix(base, "_quantized");
            if (!st.empty()) base = st;
            if (!name_matches(base, want)) continue;
            const int rc = install_by_name(g, path, ws, (int)si, base, false);
            if (rc < 0) return rc;
            if (rc == 0) break;
        }
    }
    // 2. MatMul weights stored as anonymous [K][N] constants: MatMul / MatMulInteger -> (Cast | Mul)* -> Add(bias initialiser)
    struct Found { int node; std::string bias; };
    std::vector<Found> dense_nodes;
    for (size_t ni = 0; ni < g.nodes.size(); ++ni) {
        const OnnxNode &n = g.nodes[ni];
        if ((n.op != "MatMul" && n.op != "MatMulInteger") || n.in.size() < 2 || n.out.empty() || !g.tensor(n.in[1])) continue;
        std::string cur = n.out[0], bias;
        for (int hop = 0; hop < 5 && bias.empty(); ++hop) {
            auto range = g.consumers.equal_range(cur);
            std::string next;
            for (auto it = range.first; it != range.second; ++it) {
                const OnnxNode &c = g.nodes[it->second];
                if (c.op == "Add" && c.in.size() == 2) {
                    const std::string &other = c.in[0] == cur ? c.in[1] : c.in[0];
                    const OnnxTensor *bt = g.tensor(other);
                    if (bt && bt->dims.size() == 1) { bias = other; break; }
                }
                if ((c.op == "Cast" || c.op == "Mul") && !c.out.empty()) next = c.out[0];
            }
            if (!bias.empty() || next.empty()) break;
            cur = next;
        }
        dense_nodes.push_back({(int)ni, bias});
    }
    auto install_dense = [&](const OnnxNode &n, int wslot) -> int {
        if (ws.have[wslot]) return SHODH_OK;          // already found under its HF name
        const OnnxTensor *w = g.tensor(n.in[1]);
        const TensorSlot &s = ws.slots[wslot];
        if (w->dims.size() != 2 || (uint64_t)w->dims[0] != s.cols || (uint64_t)w->dims[1] != s.rows) {
            set_error("%s: MatMul constant %s is %lld x %lld; %s needs [K = %u][N = %u]", path, w->name.c_str(), w->dims.size() > 0 ? (long long)w->dims[0] : 0, w->dims.size() > 1 ? (long long)w->dims[1] : 0, s.name.c_str(), s.cols, s.rows);
            return SHODH_ERR_INVALID;
        }
        if (n.op == "MatMul") {
            const int rc = install_by_name(g, path, ws, wslot, w->name, true);
            if (rc > 0) { set_error("%s: MatMul constant %s has an unsupported element type", path, w->name.c_str()); return SHODH_ERR_UNSUPPORTED; }
            return rc;
        }
        // MatMulInteger(a, W_q, a_zp, W_zp): scale by the quantiser's naming rule, else any float initialiser feeding the Mul chain
        std::vector<uint8_t> bytes, zp;
        std::vector<float> sc;
        if (w->external || !tensor_bytes(*w, bytes)) { set_error("%s: MatMulInteger constant %s is not uint8 / int8", path, w->name.c_str()); return SHODH_ERR_INVALID; }
        const OnnxTensor *tzp = n.in.size() > 3 && !n.in[3].empty() ? g.tensor(n.in[3]) : nullptr;
        const std::string stem = strip_suffix(w->name, "_quantized");
        const OnnxTensor *tsc = stem.empty() ? nullptr : g.tensor(stem + "_scale");
        if (!tsc) {
            std::string cur = n.out[0];
            for (int hop = 0; hop < 5 && !tsc; ++hop) {
                auto range = g.consumers.equal_range(cur);
                std::string next;
                for (auto it = range.first; it != range.second && !tsc; ++it) {
                    const OnnxNode &c = g.nodes[it->second];
                    if (c.op == "Mul") {
                        for (auto &in : c.in) {
                            if (in == cur) continue;
                            if (const OnnxTensor *t = g.tensor(in)) { if (t->dtype == 1) tsc = t; }
                            else { auto pr = g.producer.find(in); if (pr != g.producer.end() && g.nodes[pr->second].op == "Mul") for (auto &in2 : g.nodes[pr->second].in) if (const OnnxTensor *t2 = g.tensor(in2)) if (t2->dtype == 1) tsc = t2; }
                        }
                    }
                    if ((c.op == "Cast" || c.op == "Mul") && !c.out.empty()) next = c.out[0];
                }
                if (next.empty()) break;
                cur = next;
            }
        }
        if (!tsc || !tensor_floats(*tsc, sc) || sc.empty()) { set_error("%s: no weight scale found for MatMulInteger constant %s", path, w->name.c_str()); return SHODH_ERR_INVALID; }
        if (tzp) { if (tzp->dtype != w->dtype || !tensor_bytes(*tzp, zp) || zp.size() != sc.size()) { set_error("%s: zero point of %s does not match its scale / type", path, w->name.c_str()); return SHODH_ERR_INVALID; } }
        return ws.set_quantized(wslot, bytes.data(), w->dtype == 3, true, sc.data(), tzp ? zp.data() : nullptr, (uint32_t)sc.size());
    };
    auto install_bias = [&](const std::string &bias_name, int bslot) -> int {
        if (ws.have[bslot] || bias_name.empty()) return SHODH_OK;
        const int rc = install_by_name(g, path, ws, bslot, bias_name, false);
        return rc > 0 ? SHODH_OK : rc;
    };
    bool all_named = true;
    for (auto &d : dense_nodes) {
        int wslot = -1;
        for (size_t si = 0; si + 1 < ws.slots.size(); ++si)
            if (ws.slots[si].dense && name_matches(d.bias, strip_suffix(ws.slots[si].name, ".weight") + ".bias")) { wslot = (int)si; break; }
        if (wslot < 0) { all_named = false; continue; }
        SHODH_TRY(install_dense(g.nodes[d.node], wslot));
    }
    if (!all_named) {
        // anonymous biases: graph order = execution order = query, key, value, attention output, intermediate, output per layer
        std::vector<int> dslots;
        for (size_t si = 0; si < ws.slots.size(); ++si) if (ws.slots[si].dense) dslots.push_back((int)si);
        if (dense_nodes.size() == dslots.size()) {
            for (size_t i = 0; i < dslots.size(); ++i) {
                SHODH_TRY(install_dense(g.nodes[dense_nodes[i].node], dslots[i]));
                SHODH_TRY(install_bias(dense_nodes[i].bias, dslots[i] + 1));        // a dense weight's bias is the next slot
            }
        }
    }
    return ws.check_complete();
}

}  // namespace

static int load_weight_file_impl(const char *path, const shodh_embed_cfg &cfg, WeightSet &out) {
    std::vector<unsigned char> buf;
    SHODH_TRY(slurp(path, buf));
    out.init(cfg);
    const size_t n = strlen(path);
    const bool st = n > 12 && strcmp(path + n - 12, ".safetensors") == 0;
    const bool ox = n > 5 && strcmp(path + n - 5, ".onnx") == 0;
    if (st) return load_safetensors(buf, path, out);
    if (ox) return load_onnx(buf, path, out);
    // by content: a safetensors file starts with a small little-endian length followed by '{'
    if (buf.size() > 9) { uint64_t hl; memcpy(&hl, buf.data(), 8); if (hl < buf.size() && buf[8] == '{') return load_safetensors(buf, path, out); }
    return load_onnx(buf, path, out);
}
// The entry point of every file-reading ABI call (shodh_weight_file_open, shodh_embedder_load_file, shodh_embedder_create with weights_path): no C++
// exception may cross `extern "C"` -- a file is untrusted input, and an allocation sized by it (or by a big honest model on a small host) can fail.
int load_weight_file(const char *path, const shodh_embed_cfg &cfg, WeightSet &out) {
    if (!path || !*path) { set_error("empty weights path"); return SHODH_ERR_INVALID; }
    try {
        return load_weight_file_impl(path, cfg, out);
    } catch (const std::bad_alloc &) {
        set_error("%s: out of host memory while reading the file", path); return SHODH_ERR_OOM;
    } catch (const std::exception &ex) {
        set_error("%s: malformed file (%s)", path, ex.what()); return SHODH_ERR_IO;
    }
}

}  // namespace shodh

using namespace shodh;

struct shodh_weight_file { WeightSet ws; };

extern "C" {

int shodh_weight_file_open(const char *path, const shodh_embed_cfg *cfg, shodh_weight_file **out) {
    if (!path || !cfg || !out) { set_error("null argument"); return SHODH_ERR_INVALID; }
    *out = nullptr;
    shodh_weight_file *f = new shodh_weight_file();
    const int rc = load_weight_file(path, *cfg, f->ws);
    if (rc != SHODH_OK) { delete f; return rc; }
    *out = f;
    return SHODH_OK;
}
void shodh_weight_file_close(shodh_weight_file *f) { delete f; }

int shodh_weight_file_blob(const shodh_weight_file *f, float *blob_out, uint64_t n_floats) {
    if (!f || !blob_out) { set_error("null argument"); return SHODH_ERR_INVALID; }
    if (n_floats != f->ws.blob.size()) { set_error("blob_out has %llu floats, expected %llu", (unsigned long long)n_floats, (unsigned long long)f->ws.blob.size()); return SHODH_ERR_INVALID; }
    memcpy(blob_out, f->ws.blob.data(), n_floats * 4);
    return SHODH_OK;
}

int shodh_weight_file_quantized(const shodh_weight_file *f, const char *name, int8_t *q_out, uint64_t q_len, float *scale_out, int32_t *zero_point_out,
                                uint32_t scale_cap, uint32_t *n_scale_out) {
    if (!f || !name) { set_error("null argument"); return SHODH_ERR_INVALID; }
    const int si = f->ws.find(name);
    if (si < 0) { set_error("unknown tensor %s", name); return SHODH_ERR_INVALID; }
    const QTensor &t = f->ws.q[si];
    if (n_scale_out) *n_scale_out = t.present ? t.n_scale : 0;
    if (!t.present) return SHODH_OK;              // stored as floats in this file
    if (q_out) { if (q_len != t.q.size()) { set_error("%s: q_out holds %llu bytes, expected %llu", name, (unsigned long long)q_len, (unsigned long long)t.q.size()); return SHODH_ERR_INVALID; } memcpy(q_out, t.q.data(), q_len); }
    if (scale_out || zero_point_out) {
        if (scale_cap < t.n_scale) { set_error("%s: %u scales, room for %u", name, t.n_scale, scale_cap); return SHODH_ERR_INVALID; }
        if (scale_out) memcpy(scale_out, t.scale.data(), (size_t)t.n_scale * 4);
        if (zero_point_out) memcpy(zero_point_out, t.zp.data(), (size_t)t.n_scale * 4);
    }
    return SHODH_OK;
}

}  // extern "C"
