// encoder_int8.h -- the dynamically quantised dense layer of the INT8 encoder mode (included by encoder.hip only).
//
// The reference's default model is the ONNX Runtime dynamic-quantisation export `model_quint8_avx2.onnx`
// (embeddings/downloader.rs:31, minilm.rs:212-220). Its dense layers are
//     DynamicQuantizeLinear(x) -> MatMulInteger(x_q, W_q) -> float(acc) * (a_scale * w_scale) + bias
// with the activation range taken over the WHOLE input tensor, padding included (minilm.rs:588-593), 8-bit per-tensor weights,
// and everything else (softmax, GELU, LayerNorm) in fp32. This file is that layer on the matrix cores:
//   minmax_kernel      global min / max of the activation tensor (order-preserving uint keys, one atomic pair per workgroup)
//   act_quant_kernel   scale / zero point exactly as DynamicQuantizeLinear defines them, q = clip(rint(x / scale) + zp, 0, 255),
//                      stored as the SIGNED value q - 128 (v_mfma_i32_32x32x32_i8 multiplies signed bytes)
//   gemm_i8_kernel     128 x 128 x 128 tiles, 4 waves x (2 x 2) MFMA blocks, int32 accumulators (exact), epilogue
//                      acc + (128 - zp) * rowsum(W_q)  ==  sum_k (a - zp) * w   ->  float * (a_scale * w_scale[n]) + bias [+ GELU | + residual]
// Parity with the real checkpoint is unpinned (no ONNX Runtime, no weights offline); tests/test_encoder_int8_gpu.py checks the
// int32 accumulators bit for bit against a numpy restatement of these operator semantics.
#pragma once
#include "common.h"

namespace shodh {

typedef int i32x4v __attribute__((ext_vector_type(4)));
typedef int i32x16v __attribute__((ext_vector_type(16)));

// ---- weights: per-tensor symmetric 8 bit (scale = 2 max|w| / 255, zero point 128 in uint8 terms) --------------------------
__global__ __launch_bounds__(256) void absmax_kernel(const float *__restrict__ w, size_t n, uint32_t *__restrict__ out) {
    float m = 0.0f;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) m = fmaxf(m, __builtin_fabsf(w[i]));
    for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o));
    if ((threadIdx.x & 63) == 0) atomicMax(out, __float_as_uint(m));      // non-negative floats order like their bit patterns
}
__device__ __forceinline__ float weight_scale_from_absmax(float absmax) { return absmax > 0.0f ? 2.0f * absmax / 255.0f : 1.0f; }
__global__ __launch_bounds__(256) void quantize_weight_kernel(const float *__restrict__ w, size_t n, const uint32_t *__restrict__ absmax_bits,
                                                              int8_t *__restrict__ out, float *__restrict__ scale_out, int n_scale) {
    const float scale = weight_scale_from_absmax(__uint_as_float(*absmax_bits));
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
        float q = __builtin_rintf(w[i] / scale);
        q = fminf(fmaxf(q, -128.0f), 127.0f);
        out[i] = (int8_t)(int)q;
    }
    if (blockIdx.x == 0 && (int)threadIdx.x < n_scale) scale_out[threadIdx.x] = scale;    // n_scale copies: one per output feature of this matrix
}
// rowsum[n] = sum_k W_q[n][k]  (the zero-point correction term of MatMulInteger)
__global__ __launch_bounds__(64) void rowsum_s8_kernel(const int8_t *__restrict__ wq, int K, int32_t *__restrict__ rowsum) {
    const int n = blockIdx.x;
    int s = 0;
    for (int k = threadIdx.x; k < K; k += 64) s += (int)wq[(size_t)n * K + k];
    for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o);
    if (threadIdx.x == 0) rowsum[n] = s;
}

// ---- activations: DynamicQuantizeLinear ---------------------------------------------------------------------------------------
// mm[0] = min key, mm[1] = max key (order_key: ascending uint <=> ascending float); reset by act_quant_kernel's last reader? No:
// the caller memsets {0xFFFFFFFF, 0} before each use (one 8-byte memset node in the stream).
__global__ __launch_bounds__(256) void minmax_kernel(const float *__restrict__ x, size_t n, uint32_t *__restrict__ mm) {
    uint32_t lo = 0xFFFFFFFFu, hi = 0u;
    const size_t n4 = n >> 2;
    const float4 *x4 = reinterpret_cast<const float4 *>(x);
    const size_t stride = (size_t)gridDim.x * 256;
    size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    auto fold = [&](const float4 &v) {
        const uint32_t k0 = order_key(v.x), k1 = order_key(v.y), k2 = order_key(v.z), k3 = order_key(v.w);
        lo = min(lo, min(min(k0, k1), min(k2, k3)));
        hi = max(hi, max(max(k0, k1), max(k2, k3)));
    };
    for (; i + 3 * stride < n4; i += 4 * stride) {          // four 16-byte loads in flight per thread
        const float4 a = x4[i], b = x4[i + stride], c = x4[i + 2 * stride], d = x4[i + 3 * stride];
        fold(a); fold(b); fold(c); fold(d);
    }
    for (; i < n4; i += stride) fold(x4[i]);
    for (size_t i = (n4 << 2) + (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) { const uint32_t k = order_key(x[i]); lo = min(lo, k); hi = max(hi, k); }
    for (int o = 32; o > 0; o >>= 1) { lo = min(lo, (uint32_t)__shfl_xor((int)lo, o)); hi = max(hi, (uint32_t)__shfl_xor((int)hi, o)); }
    __shared__ uint32_t slo[4], shi[4];
    if ((threadIdx.x & 63) == 0) { slo[threadIdx.x >> 6] = lo; shi[threadIdx.x >> 6] = hi; }
    __syncthreads();
    if (threadIdx.x == 0) {
        atomicMin(mm + 0, min(min(slo[0], slo[1]), min(slo[2], slo[3])));
        atomicMax(mm + 1, max(max(shi[0], shi[1]), max(shi[2], shi[3])));
    }
}
struct ActQ { float scale; int zp; };
__device__ __forceinline__ ActQ act_params(const uint32_t *mm) {
    float rmin = order_key_inv(mm[0]), rmax = order_key_inv(mm[1]);
    rmin = fminf(rmin, 0.0f); rmax = fmaxf(rmax, 0.0f);
    ActQ p;
    p.scale = (rmax == rmin) ? 1.0f : (rmax - rmin) / 255.0f;
    float z = 0.0f - rmin / p.scale;
    z = fminf(fmaxf(z, 0.0f), 255.0f);
    p.zp = (int)__builtin_rintf(z);          // round half to even
    return p;
}
// q = clip(rint(x / scale) + zp, 0, 255) - 128 (signed storage); params_out = {scale, zp} for the GEMM epilogue
__global__ __launch_bounds__(256) void act_quant_kernel(const float *__restrict__ x, size_t n, const uint32_t *__restrict__ mm,
                                                        int8_t *__restrict__ out, float *__restrict__ params_out) {
    const ActQ p = act_params(mm);
    const float zpf = (float)p.zp;
    const size_t n4 = n >> 2;
    const float4 *x4 = reinterpret_cast<const float4 *>(x);
    const size_t stride = (size_t)gridDim.x * 256;
    size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    auto quant4 = [&](const float4 &v) -> uint32_t {
        const float e[4] = {v.x, v.y, v.z, v.w};
        uint32_t pk = 0;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            float q = __builtin_rintf(e[j] / p.scale) + zpf;
            q = fminf(fmaxf(q, 0.0f), 255.0f);
            pk |= (uint32_t)(((int)q - 128) & 0xFF) << (8 * j);
        }
        return pk;
    };
    uint32_t *out4 = reinterpret_cast<uint32_t *>(out);
    for (; i + 3 * stride < n4; i += 4 * stride) {          // four 16-byte loads in flight per thread
        const float4 a = x4[i], b = x4[i + stride], c = x4[i + 2 * stride], d = x4[i + 3 * stride];
        out4[i] = quant4(a); out4[i + stride] = quant4(b); out4[i + 2 * stride] = quant4(c); out4[i + 3 * stride] = quant4(d);
    }
    for (; i < n4; i += stride) out4[i] = quant4(x4[i]);
    for (size_t i = (n4 << 2) + (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
        float q = __builtin_rintf(x[i] / p.scale) + zpf;
        q = fminf(fmaxf(q, 0.0f), 255.0f);
        out[i] = (int8_t)((int)q - 128);
    }
    if (blockIdx.x == 0 && threadIdx.x == 0) { params_out[0] = p.scale; params_out[1] = zpf; }
}

// ---- int8 MFMA GEMM: acc[M,N] = Aq[M,K] (signed a - 128) * Wq[N,K]^T, then the MatMulIntegerToFloat epilogue ------------------
// Same tile structure as gemm_bf16_kernel (rows of 128 BYTES in LDS, 16-B chunks XOR-swizzled by (row>>1)&7, C computed
// transposed so that a lane owns one token and four consecutive registers are four consecutive features): a K-tile is 128
// int8 = 4 k-steps of v_mfma_i32_32x32x32_i8, whose A/B fragments are 16 bytes per lane (row = lane&31, the 16 bytes of
// k-half lane>>5). Both operands are loaded with the same (row, k-half) -> bytes rule, so the pairing of k indices is
// right whatever order the instruction walks them in. K % 128 == 0, N % 128 == 0.
enum { EPI8_BIAS = 0, EPI8_BIAS_GELU = 1, EPI8_BIAS_RESID = 2 };
__device__ __forceinline__ float gelu_erf_exact(float x) { return 0.5f * x * (1.0f + erff(x * 0.70710678118654752440f)); }
template <int EPI>
__global__ __launch_bounds__(256) void gemm_i8_kernel(const int8_t *__restrict__ A, const int8_t *__restrict__ W, const float *__restrict__ act_params /* {scale, zp} */,
                                                      const float *__restrict__ wscale /*[N]*/, const int32_t *__restrict__ rowsum /*[N]: rowsum_w[n] - K * zw[n]*/,
                                                      const float *__restrict__ bias /*[N] or null*/, const float *__restrict__ resid /*[M][N]*/,
                                                      float *__restrict__ out, int32_t *__restrict__ acc_out /* [M][N] or null */, int M, int N, int K,
                                                      uint32_t *__restrict__ mm = nullptr /* min / max keys of the output (the next layer's DynamicQuantizeLinear) */,
                                                      const int32_t *__restrict__ zw = nullptr /*[N] weight zero points in signed-storage terms, or null = all 0*/,
                                                      const int32_t *__restrict__ rsA = nullptr /*[M] row sums of the stored activations (needed with zw)*/) {
    constexpr int TB = 128 * 128;
    __shared__ __attribute__((aligned(16))) unsigned char lds[2 * 2 * TB];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int hi = lane >> 5, l31 = lane & 31;
    const int wr = wave >> 1, wc = wave & 1;
    const int m0 = blockIdx.y * 128, n0 = blockIdx.x * 128;
    typedef uint32_t u32x4q __attribute__((ext_vector_type(4)));
    u32x4q pa[4], pb[4];
    const int8_t *ga[4], *gb[4];
    int loff[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int S = i * 256 + tid, row = S >> 3, c = S & 7;
        int ra = m0 + row; if (ra >= M) ra = M - 1;
        ga[i] = A + (size_t)ra * K + c * 16;
        gb[i] = W + (size_t)(n0 + row) * K + c * 16;
        loff[i] = row * 128 + ((c ^ ((row >> 1) & 7)) << 4);
    }
    auto gload = [&](int kt) {
#pragma unroll
        for (int i = 0; i < 4; ++i) { pa[i] = *reinterpret_cast<const u32x4q *>(ga[i] + kt * 128); pb[i] = *reinterpret_cast<const u32x4q *>(gb[i] + kt * 128); }
    };
    auto stage = [&](int buf) {
        unsigned char *base = lds + buf * (2 * TB);
#pragma unroll
        for (int i = 0; i < 4; ++i) { *reinterpret_cast<u32x4q *>(base + loff[i]) = pa[i]; *reinterpret_cast<u32x4q *>(base + TB + loff[i]) = pb[i]; }
    };
    i32x16v acc[2][2];      // [n block j][m block i]
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[j][i][r] = 0;
    int fo_a[2], fo_b[2], sw_a[2], sw_b[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int ra = wr * 64 + i * 32 + l31, rb = wc * 64 + i * 32 + l31;
        fo_a[i] = ra * 128; sw_a[i] = (ra >> 1) & 7;
        fo_b[i] = TB + rb * 128; sw_b[i] = (rb >> 1) & 7;
    }
    const int nkt = K / 128;
    gload(0);
    stage(0);
    __syncthreads();
    for (int kt = 0; kt < nkt; ++kt) {
        const int cur = kt & 1;
        if (kt + 1 < nkt) gload(kt + 1);
        const unsigned char *tb = lds + cur * (2 * TB);
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            i32x4v fa[2], fb[2];
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                fa[i] = *reinterpret_cast<const i32x4v *>(tb + fo_a[i] + (((ks * 2 + hi) ^ sw_a[i]) << 4));
                fb[i] = *reinterpret_cast<const i32x4v *>(tb + fo_b[i] + (((ks * 2 + hi) ^ sw_b[i]) << 4));
            }
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int i = 0; i < 2; ++i) acc[j][i] = __builtin_amdgcn_mfma_i32_32x32x32_i8(fb[j], fa[i], acc[j][i], 0, 0, 0);
        }
        if (kt + 1 < nkt) stage(cur ^ 1);
        __syncthreads();
    }
    const float a_scale = act_params[0];
    const int corr = 128 - (int)act_params[1];        // the stored activation is a - 128: sum (a - zp) w = sum (a - 128) w + (128 - zp) rowsum(w)
    uint32_t klo = 0xFFFFFFFFu, khi = 0u;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int m = m0 + wr * 64 + i * 32 + l31;
        if (m >= M) continue;
        const int rsa = (zw && rsA) ? rsA[m] : 0;
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int nb = n0 + wc * 64 + j * 32 + 8 * g + 4 * hi;
                const float4 ws = *reinterpret_cast<const float4 *>(wscale + nb);
                const int4 rs = *reinterpret_cast<const int4 *>(rowsum + nb);
                int4 zz = make_int4(0, 0, 0, 0);
                if (zw) zz = *reinterpret_cast<const int4 *>(zw + nb);
                const float wsv[4] = {ws.x, ws.y, ws.z, ws.w};
                const int rsv[4] = {rs.x, rs.y, rs.z, rs.w};
                const int zwv[4] = {zz.x, zz.y, zz.z, zz.w};
                float bv[4] = {0.0f, 0.0f, 0.0f, 0.0f};
                if (bias) { const float4 b4 = *reinterpret_cast<const float4 *>(bias + nb); bv[0] = b4.x; bv[1] = b4.y; bv[2] = b4.z; bv[3] = b4.w; }
                float v[4];
                int ai[4];
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    ai[e] = acc[j][i][4 * g + e] + corr * rsv[e] - zwv[e] * rsa;      // sum (a - a_zp)(w' - z) = sum a'w' + c (rowsum_w - K z) - z rowsum_a
                    float x = (float)ai[e] * (a_scale * wsv[e]) + bv[e];       // MatMulIntegerToFloat: float(acc) * (a_scale * b_scale) + bias
                    if (EPI == EPI8_BIAS_GELU) x = gelu_erf_exact(x);
                    v[e] = x;
                }
                if (EPI == EPI8_BIAS_RESID) {
                    const float4 r4 = *reinterpret_cast<const float4 *>(resid + (size_t)m * N + nb);
                    v[0] += r4.x; v[1] += r4.y; v[2] += r4.z; v[3] += r4.w;
                }
                *reinterpret_cast<float4 *>(out + (size_t)m * N + nb) = make_float4(v[0], v[1], v[2], v[3]);
                if (mm) {
#pragma unroll
                    for (int e = 0; e < 4; ++e) { const uint32_t kk = order_key(v[e]); klo = min(klo, kk); khi = max(khi, kk); }
                }
                if (acc_out) *reinterpret_cast<int4 *>(acc_out + (size_t)m * N + nb) = make_int4(ai[0], ai[1], ai[2], ai[3]);
            }
    }
    if (mm) {          // (the rows loop may have skipped lanes past M: they carry the neutral keys)
        for (int o = 32; o > 0; o >>= 1) { klo = min(klo, (uint32_t)__shfl_xor((int)klo, o)); khi = max(khi, (uint32_t)__shfl_xor((int)khi, o)); }
        if (lane == 0) {
            if (klo < __atomic_load_n(mm, __ATOMIC_RELAXED)) atomicMin(mm, klo);
            if (khi > __atomic_load_n(mm + 1, __ATOMIC_RELAXED)) atomicMax(mm + 1, khi);
        }
    }
}

// one quantised weight matrix on the device
// q: row-major [N][K] signed storage; qp: the same bytes fragment-major (pack_i8_frag_kernel); scale[N]; rowsum[N] = sum_k q; zw[N] = zero points
// in signed-storage terms (null when all zero: the symmetric case); rsz[N] = rowsum - K * zw; from_export: the bytes are a file's own
struct QWeight { int8_t *q = nullptr, *qp = nullptr; float *scale = nullptr; int32_t *rowsum = nullptr, *rsz = nullptr, *zw = nullptr /* = zw_buf when any entry is non-zero */, *zw_buf = nullptr; int N = 0, K = 0; bool from_export = false;
                 int64_t zw_bound = 0; /* max over the installed rows of 128 sum_k |q| + 128 |rsz| + 128 K |zw|: below 2^24 every partial sum of the zero-point epilogue is an exactly representable float (i8_stream_gelu_kernel<., 2, .>) */ };

// quantises rows [0, N) of `w` (f32 [N][K], device) into dst rows starting at row `row0` (a fused matrix may hold several tensors,
// each with its own per-tensor scale: the scale array has one entry per output feature)
static int quantize_weight_into(const float *w, int N, int K, int8_t *q_dst, float *scale_dst, int32_t *rowsum_dst, uint32_t *scratch_u32, hipStream_t st) {
    SHODH_HIP_TRY(hipMemsetAsync(scratch_u32, 0, 4, st));
    const size_t n = (size_t)N * K;
    const uint32_t blocks = (uint32_t)std::min<size_t>(ceil_div(n, 1024), 1024);
    hipLaunchKernelGGL(absmax_kernel, dim3(blocks), dim3(256), 0, st, w, n, scratch_u32);
    // scale_dst gets N copies (<= 256 per launch of block 0: loop in chunks)
    for (int off = 0; off < N; off += 256) {
        const int cnt = std::min(256, N - off);
        hipLaunchKernelGGL(quantize_weight_kernel, dim3(off == 0 ? blocks : 1), dim3(256), 0, st, w, off == 0 ? n : 0, scratch_u32, q_dst, scale_dst + off, cnt);
    }
    hipLaunchKernelGGL(rowsum_s8_kernel, dim3((uint32_t)N), dim3(64), 0, st, q_dst, K, rowsum_dst);
    SHODH_HIP_TRY(hipGetLastError());
    return SHODH_OK;
}

// DynamicQuantizeLinear of x[n] into xq (signed storage) + params {scale, zp}; mm = 2 u32 of scratch
static int dynamic_quantize(const float *x, size_t n, int8_t *xq, float *params, uint32_t *mm, hipStream_t st) {
    SHODH_HIP_TRY(hipMemsetAsync(mm, 0xFF, 4, st));        // min key
    SHODH_HIP_TRY(hipMemsetAsync(mm + 1, 0, 4, st));       // max key
    const uint32_t blocks = (uint32_t)std::min<size_t>(std::max<size_t>(ceil_div(n, 4096), 1), 2048);
    hipLaunchKernelGGL(minmax_kernel, dim3(blocks), dim3(256), 0, st, x, n, mm);
    hipLaunchKernelGGL(act_quant_kernel, dim3(blocks), dim3(256), 0, st, x, n, mm, xq, params);
    SHODH_HIP_TRY(hipGetLastError());
    return SHODH_OK;
}

// the min / max keys are already in mm (folded into the producer's stores): only the byte pass
static int quantize_known_range(const float *x, size_t n, int8_t *xq, float *params, const uint32_t *mm, hipStream_t st) {
    const uint32_t blocks = (uint32_t)std::min<size_t>(std::max<size_t>(ceil_div(n, 4096), 1), 2048);
    hipLaunchKernelGGL(act_quant_kernel, dim3(blocks), dim3(256), 0, st, x, n, mm, xq, params);
    SHODH_HIP_TRY(hipGetLastError());
    return SHODH_OK;
}
static int reset_range(uint32_t *mm, hipStream_t st) {
    SHODH_HIP_TRY(hipMemsetAsync(mm, 0xFF, 4, st));        // min key
    SHODH_HIP_TRY(hipMemsetAsync(mm + 1, 0, 4, st));       // max key
    return SHODH_OK;
}

template <int EPI>
static int gemm_i8(const int8_t *A, const QWeight &W, int row0, int N, const float *act_params, const float *bias, const float *resid,
                   float *out, int32_t *acc_out, int M, hipStream_t st, uint32_t *mm = nullptr, const int32_t *rsA = nullptr) {
    dim3 grid(N / 128, (M + 127) / 128);
    hipLaunchKernelGGL((gemm_i8_kernel<EPI>), grid, dim3(256), 0, st, A, W.q + (size_t)row0 * W.K, act_params, W.scale + row0, W.rsz + row0, bias, resid, out, acc_out, M, N, W.K, mm,
                       W.zw ? W.zw + row0 : (const int32_t *)nullptr, rsA);
    SHODH_HIP_TRY(hipGetLastError());
    return SHODH_OK;
}

}  // namespace shodh
