// weights_io.h -- host-side model of "the encoder's parameters as a file hands them over" (shared by weights_io.hip and encoder.hip).
//
// The reference loads `model_quantized.onnx` / `model.onnx` through ONNX Runtime (minilm.rs:212-220, downloader.rs:29-53: the
// files are onnx/model_quint8_avx2.onnx and onnx/model.onnx of sentence-transformers/all-MiniLM-L6-v2). A WeightSet is what this
// library needs from such a file: every parameter of the BERT graph by its HF `BertModel` name, as f32, plus -- for the tensors
// a dynamic-quantisation export stores in 8 bits (the six dense weights per layer and the word table) -- the export's OWN bytes,
// scales and zero points, so that SHODH_DTYPE_INT8 multiplies exactly the integers ONNX Runtime would.
#pragma once
#include <cstdint>
#include <string>
#include <vector>

#include "../../include/shodh_hip.h"

namespace shodh {

// one parameter of the network, in blob order (HF BertModel order without the pooler; embedder.blob_to_state_dict names the same slices)
struct TensorSlot {
    std::string name;       // "encoder.layer.0.attention.self.query.weight", ...
    size_t offset = 0;      // into the f32 blob
    uint32_t rows = 0, cols = 0;   // [rows][cols]; 1-D tensors: rows = 1
    bool dense = false;     // a MatMul weight ([N][K] in HF layout): may arrive quantised and / or transposed
    bool quantisable = false;      // dense weights and the word table
};
std::vector<TensorSlot> tensor_table(const shodh_embed_cfg &cfg, uint64_t *n_params);

// a tensor as a quantised export stores it, converted to this library's storage convention:
//   q[n][k]   signed 8-bit. uint8 sources are stored minus 128 (the matrix cores multiply signed bytes), int8 sources as they are
//   zp[c]     the zero point in the same signed terms (uint8 zero point - 128, or the int8 zero point)
//   scale[c]  c = 0 (per tensor, n_scale == 1) or the output feature n (per channel, n_scale == N)
// real value of element (n, k) = (q[n][k] - zp[c]) * scale[c]        (DequantizeLinear / MatMulInteger semantics)
struct QTensor {
    bool present = false;
    uint32_t N = 0, K = 0, n_scale = 0;
    std::vector<int8_t> q;
    std::vector<float> scale;
    std::vector<int32_t> zp;
};

struct WeightSet {
    shodh_embed_cfg cfg{};
    std::vector<TensorSlot> slots;
    std::vector<float> blob;           // n_params f32; quantised tensors are present here DEQUANTISED (fp32 / bf16 modes use them)
    std::vector<uint8_t> have;         // per slot: 0 absent, 1 f32, 2 quantised
    std::vector<QTensor> q;            // per slot
    void init(const shodh_embed_cfg &c);
    int find(const char *name) const;  // exact HF name -> slot index, or -1
    // `data` [rows][cols] f32 (or [cols][rows] when transposed)
    int set_f32(int slot, const float *data, uint64_t n, bool transposed);
    // `data` uint8 or int8 [N][K] (or [K][N] when transposed); scale / zero_point hold n_scale entries (1 or N); zero_point may be null (= 0)
    int set_quantized(int slot, const void *data, bool is_signed, bool transposed, const float *scale, const void *zero_point, uint32_t n_scale);
    int check_complete() const;        // SHODH_OK, or SHODH_ERR_STATE naming the first missing tensor
};

// .safetensors (F32 / F16 / BF16 tensors under HF names, any "prefix." in front) or .onnx (f32 export or dynamic-quantisation export)
int load_weight_file(const char *path, const shodh_embed_cfg &cfg, WeightSet &out);

}  // namespace shodh
