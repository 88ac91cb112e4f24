// encoder_int8_fast.h -- the INT8 encoder layer without f32 round trips (round 3; included by encoder.hip only).
//
// Same function as encoder_int8.h's path (the dynamic-quantisation export's graph, minilm.rs:212-220, :588-593): every constant-weight
// MatMul is DynamicQuantizeLinear(whole tensor) -> MatMulInteger -> float(acc) * (a_scale * w_scale[n]) + bias, everything else fp32.
// What changed is where the tensors live. Round 2 wrote QKV (4.8 GB at 4096 padded texts), the GELU output (6.4 GB) and two pre-norm
// sums (1.6 GB each) to HBM in f32 and read them back, per layer; here
//   qkv_attn_i8_kernel     one workgroup per (sequence, head): Q/K/V projections on v_mfma_i32_32x32x32_i8 straight into the MFMA
//                          operand layouts of the attention (K and V^T fragments through LDS, Q and P in registers), softmax in f32,
//                          both attention products as three f16 MFMAs on (hi, lo) splits of the f32 operands (error <= 2^-21 relative,
//                          see f16_split). K and V are projected for the real keys only -- masked keys are never read.
//   qkv_attn_seq_kernel    the same per SEQUENCE (the default): the f32 layer input is fetched once, quantised in the kernel and kept as
//                          fragments for all twelve heads; head weights through LDS by LDS-DMA; softmax in one piece over <= 4 key blocks.
//   i8_stream_gelu_kernel  the two FFN-up passes (range, then bytes) of i8_stream_kernel with the epilogue of one token block running
//                          under the MFMAs of the next (the default for those two passes).
//   i8_stream_kernel       weight-stationary int8 GEMM for K = 384 (attention output, FFN up): 12 waves hold 384 output features as
//                          resident fragments, the quantised activations stream through LDS by LDS-DMA (gemm_k384_stream_kernel's
//                          structure). Epilogues: + residual + LayerNorm (whole rows in the workgroup) | GELU -> range only |
//                          GELU -> quantised bytes (the range is known from the first pass; the f32 intermediate never exists).
//   i8_ktile_ln_kernel     tiled int8 GEMM for K = 1536 (FFN down) over all 384 features + residual + LayerNorm.
// Weight zero points are general (per tensor or per output channel, any value): with a' = a - 128, w' the signed storage and
// z[n] its zero point, sum_k (a - a_zp)(w' - z[n]) = sum a'w' + c * (rowsum_w[n] - K z[n]) - z[n] * rowsum_a[m],  c = 128 - a_zp.
#pragma once
#include <type_traits>
#include "common.h"
#include "glds.h"
#include "encoder_int8.h"

namespace shodh {

typedef _Float16 f16x8q __attribute__((ext_vector_type(8)));
typedef float f32x16q __attribute__((ext_vector_type(16)));
typedef float f32x4q __attribute__((ext_vector_type(4)));
typedef int i32x4q __attribute__((ext_vector_type(4)));
typedef uint32_t u32x4qq __attribute__((ext_vector_type(4)));
typedef int i32x16l __attribute__((ext_vector_type(16)));
#include <utility>
// f(integral_constant<int, 0>), f(integral_constant<int, 1>), ...: a loop whose index is a constant expression in the body (immediate operands of s_waitcnt)
template <int... I, class F> __device__ __forceinline__ void static_for_i(std::integer_sequence<int, I...>, F &&f) { (f(std::integral_constant<int, I>{}), ...); }
// LDS hand-over between the waves of a workgroup WITHOUT touching vmcnt: __syncthreads() would also wait for the LDS-DMA in flight
__device__ __forceinline__ void lds_barrier() {
    __builtin_amdgcn_s_waitcnt(0xC07F);   // lgkmcnt(0)
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
}

// GELU(x) = x Phi(x) = max(x, 0) - (|x| / 2) erfc(|x| / sqrt 2), with erfc(z) = 2^P(z) on z in [0, 4.4]: P = the degree-8 weighted least-squares fit of
// log2 erfc (Chebyshev nodes, weight erfc: what counts is the absolute error of erfc), fitted on [0, 4.4] (|x| <= 6.2; erfc(4.4) = 5e-10).
// Measured in f32 Horner arithmetic over x in [-8, 8] against the exact function: |erfc error| <= 8.1e-8, |gelu error| <= 3.0e-7 absolute and
// <= 9.5e-8 |x| (+ one ulp of v_exp_f32) -- tighter than Abramowitz-Stegun 7.1.26 (1.5e-7 on erf), which this replaces: 7.1.26 needs v_rcp AND v_exp
// (quarter-rate instructions, 8 issue slots per value) plus 8 more slots; this needs one v_exp and eight fused multiply-adds on the packed-FP32 pipe
// (v_pk_fma_f32: two values per slot): ~12 slots per value against ~16. Exact erff costs ~40. The INT8 FFN evaluates it on 1.6e9 values per layer.
typedef float f32x2q __attribute__((ext_vector_type(2)));
__device__ __forceinline__ f32x2q gelu_i8x2(f32x2q x) {
    // gelu(x) = max(x, 0) - |x| / 2 * erfc(|x| / sqrt 2), erfc(z) = 2^P(z). No clamp on z: past the fitted interval (z > 4.4, |x| > 6.2) P keeps
    // falling faster than log2 erfc does (P(5) = -39.5, P(8) = -171, P(20) = -3.8e5, -inf from 1e10 on: tests/test_gelu_poly_cpu.py), so the
    // correction term only shrinks below its 5e-10 |x| there. |x| itself is never formed: z and |x| / 2 come from multiplies with the |.| operand
    // modifier.
    f32x2q z; z.x = __builtin_fabsf(x.x) * 0.70710678118654752440f; z.y = __builtin_fabsf(x.y) * 0.70710678118654752440f;
    f32x2q p = __builtin_elementwise_fma(z, (f32x2q)-2.753492845e-05f, (f32x2q)3.102343180e-04f);
    p = __builtin_elementwise_fma(p, z, (f32x2q)-1.085499767e-03f);
    p = __builtin_elementwise_fma(p, z, (f32x2q)-1.382132061e-03f);
    p = __builtin_elementwise_fma(p, z, (f32x2q)2.874283120e-02f);
    p = __builtin_elementwise_fma(p, z, (f32x2q)-1.486884505e-01f);
    p = __builtin_elementwise_fma(p, z, (f32x2q)-9.183745980e-01f);
    p = __builtin_elementwise_fma(p, z, (f32x2q)-1.627911806e+00f);
    p = __builtin_elementwise_fma(p, z, (f32x2q)4.870492631e-08f);
    f32x2q e; e.x = __builtin_amdgcn_exp2f(p.x); e.y = __builtin_amdgcn_exp2f(p.y);      // erfc(z)
    f32x2q m; m.x = __builtin_fabsf(x.x) * 0.5f; m.y = __builtin_fabsf(x.y) * 0.5f;
    f32x2q r; r.x = fmaxf(x.x, 0.0f); r.y = fmaxf(x.y, 0.0f);
    return __builtin_elementwise_fma(-m, e, r);
}
__device__ __forceinline__ float gelu_i8(float x) { f32x2q v; v.x = x; v.y = x; return gelu_i8x2(v).x; }      // the same bits as the packed form
// f32 -> uint8 the way DynamicQuantizeLinear writes it, q = saturate(round_half_even(t) + zp), four values into one dword: v_rndne + add + ONE
// v_cvt_pk_u8_f32 per value (round-half-even and saturation to [0, 255] measured on gfx950, tools/ubench/cvt_probe.hip; `rq` is already integral,
// so the instruction's own rounding is idle). The stored byte is q - 128 = q ^ 0x80: one XOR per dword.
__device__ __forceinline__ uint32_t pack_u8(uint32_t acc, float rq, int byte) {
    uint32_t r;
    switch (byte) {
        case 0: asm("v_cvt_pk_u8_f32 %0, %1, 0, %2" : "=v"(r) : "v"(rq), "v"(acc)); break;
        case 1: asm("v_cvt_pk_u8_f32 %0, %1, 1, %2" : "=v"(r) : "v"(rq), "v"(acc)); break;
        case 2: asm("v_cvt_pk_u8_f32 %0, %1, 2, %2" : "=v"(r) : "v"(rq), "v"(acc)); break;
        default: asm("v_cvt_pk_u8_f32 %0, %1, 3, %2" : "=v"(r) : "v"(rq), "v"(acc)); break;
    }
    return r;
}
// GELU has ONE minimum, at x* = -0.75179...: decreasing on (-inf, x*], increasing on [x*, +inf). So the range of gelu over a set is
// decided by three of its elements: the largest one (maximum, if positive) and the two nearest to x* from either side (minimum).
// The range pass of the FFN tracks those three pre-activations (3 compares / selects per value instead of a GELU evaluation) and
// gelu_range_finalize_kernel evaluates gelu_i8 on them: the values the quantising pass will produce for those very elements. Exact for
// the function itself; for gelu_i8 (|error| <= 1e-7 |x|) the extremes found this way can differ from the extremes of the computed
// tensor by that approximation error at most -- a relative 1e-7 on the scale, below the approximation's own effect on the bytes.
constexpr float GELU_ARGMIN = -0.7517915964f;
// stats[0] = key of max{x <= x*} (atomicMax, 0 = none), stats[1] = key of min{x >= x*} (atomicMin, 0xFFFFFFFF = none), stats[2] = key of max x
__device__ __forceinline__ void gelu_range_from_stats(const uint32_t *stats, float &lo, float &hi) {
    lo = 0.0f; hi = 0.0f;                 // DynamicQuantizeLinear widens the range to contain 0 anyway
    if (stats[0] != 0u) { const float g = gelu_i8(order_key_inv(stats[0])); lo = fminf(lo, g); hi = fmaxf(hi, g); }
    if (stats[1] != 0xFFFFFFFFu) { const float g = gelu_i8(order_key_inv(stats[1])); lo = fminf(lo, g); hi = fmaxf(hi, g); }
    if (stats[2] != 0u) { const float g = gelu_i8(order_key_inv(stats[2])); lo = fminf(lo, g); hi = fmaxf(hi, g); }
}
// n = 1: the batch tensor; n = sequences: one range per text (SHODH_QUANT_SCOPE_PER_TEXT), stats [n][4], mm [n][2]
__global__ void gelu_range_finalize_kernel(const uint32_t *__restrict__ stats, uint32_t *__restrict__ mm, int n) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    float lo, hi;
    gelu_range_from_stats(stats + 4 * i, lo, hi);
    mm[2 * i] = order_key(lo); mm[2 * i + 1] = order_key(hi);
}

// W_q[N][K] signed bytes -> fragment-major [N/32][K/32][64 lanes][16]: lane = ((k % 32) / 16) * 32 + n % 32 holds bytes k % 16
__global__ void pack_i8_frag_kernel(const int8_t *__restrict__ W, int8_t *__restrict__ out, int N, int K) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (size_t)N * K) return;
    const int n = (int)(i / K), k = (int)(i % K);
    const size_t frag = ((size_t)(n / 32) * (K / 32) + k / 32) * 64 + ((k % 32) / 16) * 32 + n % 32;
    out[frag * 16 + k % 16] = W[i];
}
// rsz[n] = rowsum_w[n] - K * z[n]
__global__ void rsz_kernel(const int32_t *__restrict__ rowsum, const int32_t *__restrict__ zw /* or null */, int32_t *__restrict__ rsz, int N, int K) {
    const int n = blockIdx.x * blockDim.x + threadIdx.x;
    if (n < N) rsz[n] = rowsum[n] - (zw ? K * zw[n] : 0);
}

// DynamicQuantizeLinear bytes + per-row sums of the stored (signed) values: half a wave per row
__global__ __launch_bounds__(256) void act_quant_rows_kernel(const float *__restrict__ x, int M, int K, const uint32_t *__restrict__ mm,
                                                             int8_t *__restrict__ out, float *__restrict__ params_out, int32_t *__restrict__ rowsum) {
    const ActQ p = act_params(mm);
    const float zpf = (float)p.zp;
    const int l = threadIdx.x & 31;
    const int k4 = K >> 2;
    for (int row = (blockIdx.x * 256 + threadIdx.x) >> 5; row < M; row += (gridDim.x * 256) >> 5) {
        const float4 *x4 = reinterpret_cast<const float4 *>(x + (size_t)row * K);
        uint32_t *o4 = reinterpret_cast<uint32_t *>(out + (size_t)row * K);
        int s = 0;
        for (int g = l; g < k4; g += 32) {
            const float4 v = x4[g];
            const float e[4] = {v.x, v.y, v.z, v.w};
            uint32_t pk = 0;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                float q = __builtin_rintf(e[j] / p.scale) + zpf;
                q = fminf(fmaxf(q, 0.0f), 255.0f);
                const int qi = (int)q - 128;
                s += qi;
                pk |= (uint32_t)(qi & 0xFF) << (8 * j);
            }
            o4[g] = pk;
        }
        for (int o = 16; o > 0; o >>= 1) s += __shfl_xor(s, o);
        if (l == 0) rowsum[row] = s;
    }
    if (blockIdx.x == 0 && threadIdx.x == 0) { params_out[0] = p.scale; params_out[1] = zpf; }
}

// the same with one range per SEQUENCE of `rows_per_seq` rows (SHODH_QUANT_SCOPE_PER_TEXT: every DynamicQuantizeLinear spans one text's
// [max_len, K] tensor); rowsum may be null
__global__ __launch_bounds__(256) void act_quant_seq_kernel(const float *__restrict__ x, int M, int K, const uint32_t *__restrict__ mm, int rows_per_seq,
                                                            int8_t *__restrict__ out, int32_t *__restrict__ rowsum) {
    const int l = threadIdx.x & 31;
    const int k4 = K >> 2;
    for (int row = (blockIdx.x * 256 + threadIdx.x) >> 5; row < M; row += (gridDim.x * 256) >> 5) {
        const ActQ p = act_params(mm + 2 * (row / rows_per_seq));
        const float zpf = (float)p.zp;
        const float4 *x4 = reinterpret_cast<const float4 *>(x + (size_t)row * K);
        uint32_t *o4 = reinterpret_cast<uint32_t *>(out + (size_t)row * K);
        int s = 0;
        for (int g = l; g < k4; g += 32) {
            const float4 v = x4[g];
            const float e[4] = {v.x, v.y, v.z, v.w};
            uint32_t pk = 0;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                float q = __builtin_rintf(e[j] / p.scale) + zpf;
                q = fminf(fmaxf(q, 0.0f), 255.0f);
                const int qi = (int)q - 128;
                s += qi;
                pk |= (uint32_t)(qi & 0xFF) << (8 * j);
            }
            o4[g] = pk;
        }
        if (rowsum) {
            for (int o = 16; o > 0; o >>= 1) s += __shfl_xor(s, o);
            if (l == 0) rowsum[row] = s;
        }
    }
}

// x = hi + lo with hi = f16(x), lo = f16(x - hi): |x - (hi + lo)| <= 2^-22 |x| (+ 2^-25 absolute where lo is subnormal), so
// a.b ~ ah.bh + ah.bl + al.bh drops only al.bl: relative error <= 2^-21 per product, f32 accumulation in the MFMA as for any f32 dot.
// Three instructions per pair of values: v_cvt_pk_f16_f32 for the hi parts, then v_fma_mixlo_f16 / v_fma_mixhi_f16 compute x * 1.0 - hi with the hi
// part read as the f16 it is and write the f16 result: x - hi is exact in f32 (hi is x rounded to 11 bits), so the one rounding is the f16 one --
// the same values as (f16)(x - (float)hi), which costs six (two conversions back, two subtractions, one more pack). tools/ubench/mix_split_probe.hip
// compares the two forms bit for bit. ONE asm block per eight values, closed by a wait state: v_fma_mixhi_f16 writes half a register, and on this
// chip a VALU / MFMA instruction that reads such a register in the very next slot gets the old half (the compiler pads this hazard for its own
// instructions, it cannot see into an asm block -- found as one wrong text in the fixture test when the split sat right in front of an MFMA).
__device__ __forceinline__ void f16_split_asm(float a0, float a1, float a2, float a3, float a4, float a5, float a6, float a7, f16x8q &h, f16x8q &l) {
    typedef uint32_t u32x4s __attribute__((ext_vector_type(4)));
    u32x4s hh, ll;
    uint32_t h0, h1, h2, h3, l0, l1, l2, l3;
    asm("v_cvt_pk_f16_f32 %0, %8, %9\n\t"
        "v_cvt_pk_f16_f32 %1, %10, %11\n\t"
        "v_cvt_pk_f16_f32 %2, %12, %13\n\t"
        "v_cvt_pk_f16_f32 %3, %14, %15\n\t"
        "v_fma_mixlo_f16 %4, %8, 1.0, -%0 op_sel:[0,0,0] op_sel_hi:[0,0,1]\n\t"
        "v_fma_mixlo_f16 %5, %10, 1.0, -%1 op_sel:[0,0,0] op_sel_hi:[0,0,1]\n\t"
        "v_fma_mixlo_f16 %6, %12, 1.0, -%2 op_sel:[0,0,0] op_sel_hi:[0,0,1]\n\t"
        "v_fma_mixlo_f16 %7, %14, 1.0, -%3 op_sel:[0,0,0] op_sel_hi:[0,0,1]\n\t"
        "v_fma_mixhi_f16 %4, %9, 1.0, -%0 op_sel:[0,0,1] op_sel_hi:[0,0,1]\n\t"
        "v_fma_mixhi_f16 %5, %11, 1.0, -%1 op_sel:[0,0,1] op_sel_hi:[0,0,1]\n\t"
        "v_fma_mixhi_f16 %6, %13, 1.0, -%2 op_sel:[0,0,1] op_sel_hi:[0,0,1]\n\t"
        "v_fma_mixhi_f16 %7, %15, 1.0, -%3 op_sel:[0,0,1] op_sel_hi:[0,0,1]\n\t"
        "s_nop 1"
        : "=&v"(h0), "=&v"(h1), "=&v"(h2), "=&v"(h3), "=&v"(l0), "=&v"(l1), "=&v"(l2), "=&v"(l3)
        : "v"(a0), "v"(a1), "v"(a2), "v"(a3), "v"(a4), "v"(a5), "v"(a6), "v"(a7));
    hh[0] = h0; hh[1] = h1; hh[2] = h2; hh[3] = h3; ll[0] = l0; ll[1] = l1; ll[2] = l2; ll[3] = l3;
    h = __builtin_bit_cast(f16x8q, hh);
    l = __builtin_bit_cast(f16x8q, ll);
}
// (the per-head kernel, at 128 registers, keeps the plain form)
__device__ __forceinline__ void f16_split8(const float *v, f16x8q &h, f16x8q &l) {
#pragma unroll
    for (int j = 0; j < 8; ++j) { const _Float16 a = (_Float16)v[j]; h[j] = a; l[j] = (_Float16)(v[j] - (float)a); }
}
// the same for four PAIRS of values (the callers keep their arithmetic on the packed-FP32 pipe: a pair is a 64-bit register pair)
__device__ __forceinline__ void f16_split8p(const f32x2q *v2, f16x8q &h, f16x8q &l) {
    f16_split_asm(v2[0][0], v2[0][1], v2[1][0], v2[1][1], v2[2][0], v2[2][1], v2[3][0], v2[3][1], h, l);
}

// ---- fused Q/K/V projection + attention --------------------------------------------------------------------------------------
// grid: ceil(nseq / 8) * 8 * heads workgroups of 4 waves; workgroup id -> (XCD, sequence, head) so that the `heads` workgroups of a
// sequence run on ONE XCD (consecutive ids go round the 8 XCDs): its quantised rows come from HBM once and then from that L2.
// Layouts (f(r, hi) = (r & 3) + 8 (r >> 2) + 4 hi is the row that register r of half-wave hi holds in a 32x32 MFMA result):
//   Q  = Wq . X^T   A = weight fragments (rows = features), B = token fragments  -> lane = query, registers = features f(r, hi)
//                   == the B operand of S^T = K . Q^T with k-slot (hi, j) of step s bound to feature f(8 s + j, hi)
//   K  likewise (lane = key): the same registers are the A operand of S^T under the same k-slot rule -> LDS, fragment-major
//   V^T = X . Wv^T  A = token fragments, B = weight fragments -> lane = feature d, registers = keys f(r, hi) of the block
//                   == the A operand of O^T = V^T . P^T with k-slot (hi, j) of step s bound to key f(8 s + j, hi), which is the
//                   order in which S^T's result hands the keys (and so P) to a lane: P never leaves its registers.
// LDS: K / V fragments [key block][K | V][step 0..1][hi | lo part][64 lanes][16 B] = 8 KiB per 32 keys, then 4 KiB of output
// scratch per wave.
__global__ __launch_bounds__(256, 4) void qkv_attn_i8_kernel(const int8_t *__restrict__ XQ, const int32_t *__restrict__ rsA /* row sums of XQ, or null */,
                                                             const uint32_t *__restrict__ mmA /* range keys of the quantised tensor */,
                                                             const int8_t *__restrict__ Wp /* fused q|k|v weight, fragment-major */, const float *__restrict__ wscale,
                                                             const int32_t *__restrict__ rsz, const int32_t *__restrict__ zw /* or null */, const float *__restrict__ bias,
                                                             const int32_t *__restrict__ cu, const int32_t *__restrict__ klen /* or null */, float *__restrict__ ctx,
                                                             uint32_t *__restrict__ mm_out, int nseq, int heads, int H, int kv_bytes) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int xcd = blockIdx.x & 7, rr = blockIdx.x >> 3;
    const int seq = (rr / heads) * 8 + xcd, head = rr % heads;
    if (seq >= nseq) return;                              // the whole workgroup
    const int t0 = cu[seq], P = cu[seq + 1] - t0;
    const int S = klen ? klen[seq] : P;
    const int tid = threadIdx.x, lane = tid & 63, hi = lane >> 5, l31 = lane & 31;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int nkb = (S + 31) >> 5, nqb = (P + 31) >> 5;
    const ActQ ap = act_params(mmA);
    const float a_scale = ap.scale;
    const int corr = 128 - ap.zp;
    const i32x4q *wp4 = reinterpret_cast<const i32x4q *>(Wp);
    const int nblk_h = H >> 5;                            // 32-feature blocks per q / k / v section (12)
    uint32_t klo = 0xFFFFFFFFu, khi = 0u;

    // ---- phase 1: K and V^T fragments of the real keys -> LDS
    for (int task = wave; task < 2 * nkb; task += 4) {
        const int kb = task >> 1, isv = task & 1;
        int tok = kb * 32 + l31; if (tok >= P) tok = P - 1;
        const int8_t *xr = XQ + (size_t)(t0 + tok) * 384 + hi * 16;
        const i32x4q *wf = wp4 + ((size_t)((isv ? 2 : 1) * nblk_h + head) * 12) * 64 + lane;
        i32x16l acc;
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[r] = 0;
        // four k-steps (eight 16-byte loads) at a time: with four workgroups per CU the other waves cover the round trips, and the whole
        // 24-load prologue of a full unroll would not fit 128 registers
        if (isv) {
#pragma unroll 4
            for (int ks = 0; ks < 12; ++ks) acc = __builtin_amdgcn_mfma_i32_32x32x32_i8(*reinterpret_cast<const i32x4q *>(xr + ks * 32), wf[ks * 64], acc, 0, 0, 0);
        } else {
#pragma unroll 4
            for (int ks = 0; ks < 12; ++ks) acc = __builtin_amdgcn_mfma_i32_32x32x32_i8(wf[ks * 64], *reinterpret_cast<const i32x4q *>(xr + ks * 32), acc, 0, 0, 0);
        }
        float v[16];
        if (!isv) {                                       // lane = key `tok`, registers = features
            const int rsa = (rsA && zw) ? rsA[t0 + tok] : 0;
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int nb = H + head * 32 + 8 * g + 4 * hi;
                const f32x4q ws = *reinterpret_cast<const f32x4q *>(wscale + nb), b4 = *reinterpret_cast<const f32x4q *>(bias + nb);
                const i32x4q rz = *reinterpret_cast<const i32x4q *>(rsz + nb);
                i32x4q z4 = {0, 0, 0, 0};
                if (zw) z4 = *reinterpret_cast<const i32x4q *>(zw + nb);
#pragma unroll
                for (int e = 0; e < 4; ++e) v[4 * g + e] = (float)(acc[4 * g + e] + corr * rz[e] - z4[e] * rsa) * (a_scale * ws[e]) + b4[e];
            }
        } else {                                          // lane = feature d, registers = keys f(r, hi) of the block
            const int n = 2 * H + head * 32 + l31;
            const float ws1 = wscale[n], b1 = bias[n];
            const int rz1 = rsz[n], z1 = zw ? zw[n] : 0;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                int tk = kb * 32 + (r & 3) + 8 * (r >> 2) + 4 * hi; if (tk >= P) tk = P - 1;
                const int rsa = (rsA && zw) ? rsA[t0 + tk] : 0;
                v[r] = (float)(acc[r] + corr * rz1 - z1 * rsa) * (a_scale * ws1) + b1;
            }
        }
#pragma unroll
        for (int s2 = 0; s2 < 2; ++s2) {
            f16x8q fh, fl;
            f16_split8(v + 8 * s2, fh, fl);
            unsigned char *dst = smem + (size_t)((kb * 2 + isv) * 2 + s2) * 2048 + lane * 16;
            *reinterpret_cast<f16x8q *>(dst) = fh;
            *reinterpret_cast<f16x8q *>(dst + 1024) = fl;
        }
    }
    __syncthreads();

    // ---- phase 2: per 32-query block: Q projection, online softmax over the key blocks, O^T, row-wise stores
    const float sc = 0.17677669529663688110f * 1.44269504088896340736f;      // 1/sqrt(32) * log2(e): softmax in base 2
    unsigned char *scr = smem + kv_bytes + wave * 2048;       // 16 tokens x 128 B: the 32-query block leaves in two halves
    for (int qb = wave; qb < nqb; qb += 4) {
        const int q = qb * 32 + l31;
        const int qc = q < P ? q : P - 1;
        const int8_t *xr = XQ + (size_t)(t0 + qc) * 384 + hi * 16;
        const i32x4q *wf = wp4 + ((size_t)head * 12) * 64 + lane;
        i32x16l acc;
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[r] = 0;
#pragma unroll 4
        for (int ks = 0; ks < 12; ++ks) acc = __builtin_amdgcn_mfma_i32_32x32x32_i8(wf[ks * 64], *reinterpret_cast<const i32x4q *>(xr + ks * 32), acc, 0, 0, 0);
        f16x8q bqh[2], bql[2];
        {
            float v[16];
            const int rsa = (rsA && zw) ? rsA[t0 + qc] : 0;
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int nb = head * 32 + 8 * g + 4 * hi;
                const f32x4q ws = *reinterpret_cast<const f32x4q *>(wscale + nb), b4 = *reinterpret_cast<const f32x4q *>(bias + nb);
                const i32x4q rz = *reinterpret_cast<const i32x4q *>(rsz + nb);
                i32x4q z4 = {0, 0, 0, 0};
                if (zw) z4 = *reinterpret_cast<const i32x4q *>(zw + nb);
#pragma unroll
                for (int e = 0; e < 4; ++e) v[4 * g + e] = ((float)(acc[4 * g + e] + corr * rz[e] - z4[e] * rsa) * (a_scale * ws[e]) + b4[e]) * sc;
            }
            f16_split8(v, bqh[0], bql[0]);
            f16_split8(v + 8, bqh[1], bql[1]);
        }
        f32x16q o;
#pragma unroll
        for (int r = 0; r < 16; ++r) o[r] = 0.0f;
        float mx = -3.0e38f, l = 0.0f;
        for (int kb = 0; kb < nkb; ++kb) {
            f32x16q st;
#pragma unroll
            for (int r = 0; r < 16; ++r) st[r] = 0.0f;
            const unsigned char *kf = smem + (size_t)(kb * 2) * 4096 + lane * 16;
#pragma unroll
            for (int s2 = 0; s2 < 2; ++s2) {
                const f16x8q kh = *reinterpret_cast<const f16x8q *>(kf + s2 * 2048), kl = *reinterpret_cast<const f16x8q *>(kf + s2 * 2048 + 1024);
                st = __builtin_amdgcn_mfma_f32_32x32x16_f16(kh, bqh[s2], st, 0, 0, 0);
                st = __builtin_amdgcn_mfma_f32_32x32x16_f16(kh, bql[s2], st, 0, 0, 0);
                st = __builtin_amdgcn_mfma_f32_32x32x16_f16(kl, bqh[s2], st, 0, 0, 0);
            }
            float bm = -3.0e38f;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int key = kb * 32 + (r & 3) + 8 * (r >> 2) + 4 * hi;
                st[r] = key < S ? st[r] : -3.0e38f;
                bm = fmaxf(bm, st[r]);
            }
            bm = fmaxf(bm, __shfl_xor(bm, 32));
            const float mn = fmaxf(mx, bm);
            const float cf = __builtin_amdgcn_exp2f(mx - mn);
            float ps = 0.0f;
            float pv[16];
#pragma unroll
            for (int r = 0; r < 16; ++r) { pv[r] = __builtin_amdgcn_exp2f(st[r] - mn); ps += pv[r]; }      // masked keys: exp2(-3e38 - mn) = 0
            ps += __shfl_xor(ps, 32);
            l = l * cf + ps;
            mx = mn;
#pragma unroll
            for (int r = 0; r < 16; ++r) o[r] *= cf;
            const unsigned char *vf = smem + (size_t)(kb * 2 + 1) * 4096 + lane * 16;
#pragma unroll
            for (int s2 = 0; s2 < 2; ++s2) {
                f16x8q ph, pl;
                f16_split8(pv + 8 * s2, ph, pl);
                const f16x8q vh = *reinterpret_cast<const f16x8q *>(vf + s2 * 2048), vl = *reinterpret_cast<const f16x8q *>(vf + s2 * 2048 + 1024);
                o = __builtin_amdgcn_mfma_f32_32x32x16_f16(vh, ph, o, 0, 0, 0);
                o = __builtin_amdgcn_mfma_f32_32x32x16_f16(vh, pl, o, 0, 0, 0);
                o = __builtin_amdgcn_mfma_f32_32x32x16_f16(vl, ph, o, 0, 0, 0);
            }
        }
        // O^T: this lane = query q, register r = feature f(r, hi). Through the wave's scratch (16-byte chunks XOR-swizzled by
        // token & 7) so that a token's 32 features leave as one 128-byte row; two halves of 16 queries (2 KiB of scratch per wave
        // keeps four workgroups on a CU).
        const float invl = 1.0f / l;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            o[r] *= invl;
            if (q < P) { const uint32_t kk = order_key(o[r]); klo = min(klo, kk); khi = max(khi, kk); }
        }
#pragma unroll
        for (int half = 0; half < 2; ++half) {
            if ((l31 >> 4) == half) {
                const int tl = l31 & 15;
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    f32x4q ov;
#pragma unroll
                    for (int e = 0; e < 4; ++e) ov[e] = o[4 * g + e];
                    *reinterpret_cast<f32x4q *>(scr + tl * 128 + (((2 * g + hi) ^ (tl & 7)) << 4)) = ov;
                }
            }
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                const int tl = h * 8 + (lane >> 3), ch = lane & 7;
                const f32x4q v4 = *reinterpret_cast<const f32x4q *>(scr + tl * 128 + ((ch ^ (tl & 7)) << 4));
                const int qt = qb * 32 + half * 16 + tl;
                if (qt < P) *reinterpret_cast<f32x4q *>(ctx + (size_t)(t0 + qt) * H + head * 32 + ch * 4) = v4;
            }
        }
    }
    if (mm_out) {
        for (int ofs = 32; ofs > 0; ofs >>= 1) { klo = min(klo, (uint32_t)__shfl_xor((int)klo, ofs)); khi = max(khi, (uint32_t)__shfl_xor((int)khi, ofs)); }
        if (lane == 0) {
            if (klo < __atomic_load_n(mm_out, __ATOMIC_RELAXED)) atomicMin(mm_out, klo);
            if (khi > __atomic_load_n(mm_out + 1, __ATOMIC_RELAXED)) atomicMax(mm_out + 1, khi);
        }
    }
}

// half a token block (16 rows = 24 KiB of f32, contiguous at `rows`) as 24 fully coalesced 1-KiB requests per wave: lane i of request j holds
// the 16-byte piece 64 j + i of the span (row (64 j + i) / 96, piece (64 j + i) % 96 of it)
__device__ __forceinline__ void fetch_rows16(const float *rows, int lane, f32x4q (&v)[24]) {
    const f32x4q *bp = reinterpret_cast<const f32x4q *>(rows) + lane;
#pragma unroll
    for (int i = 0; i < 24; ++i) v[i] = bp[i * 64];
}
// DynamicQuantizeLinear of a fetched half block with the parameters (q_scale, q_zpf) -> row-major bytes at dst (16 rows x 384 B, 16-byte chunks
// XOR-swizzled by row & 7).
// round_half_even(x / scale) without a division per element, and still the operator's value: t = x * fl(1 / scale) and fl(x / scale) both lie
// within 1.5 * 2^-23 |t| of each other (two roundings against one), so rint(t) can differ from rint(fl(x / scale)) only when t lies that close
// to a half-integer. |t| <= 255 for every value of the tensor whose range defines the scale, so the test is |t - rint(t)| >= 1/2 - 2^-14
// (2^-22 * 256: a third more than needed) on the largest residual of a lane's sixteen values (v_max3 with |.| modifiers: half an issue slot
// per value), and a 16-value group in which any lane of the wave sees such a value is redone with the division (about one group in nine).
// The division sequence costs ~14 issue slots per value, this ~4; 192 values per lane and sequence.
__device__ __forceinline__ void quant_half16(const f32x4q (&v)[24], unsigned char *dst, float q_scale, float q_zpf, int lane) {
    const float r_scale = 1.0f / q_scale;
#pragma unroll
    for (int i4 = 0; i4 < 6; ++i4) {
        float rq[16];
        float dm = 0.0f;                                // largest |t - rint(t)| of the lane's sixteen values
#pragma unroll
        for (int c = 0; c < 4; ++c)
#pragma unroll
            for (int e2 = 0; e2 < 2; ++e2) {            // two values at a time: the multiply, the residual and the zero-point add run on the packed-FP32 pipe
                f32x2q x2; x2[0] = v[4 * i4 + c][2 * e2]; x2[1] = v[4 * i4 + c][2 * e2 + 1];
                const f32x2q t2 = x2 * r_scale;
                f32x2q q2; q2[0] = __builtin_rintf(t2[0]); q2[1] = __builtin_rintf(t2[1]);
                const f32x2q d2 = t2 - q2;
                dm = __builtin_fmaxf(__builtin_fmaxf(dm, __builtin_fabsf(d2[0])), __builtin_fabsf(d2[1]));
                const f32x2q r2 = q2 + q_zpf;
                rq[4 * c + 2 * e2] = r2[0]; rq[4 * c + 2 * e2 + 1] = r2[1];
            }
        const bool amb = dm >= 0.5f - 6.103515625e-05f;    // within 2^-22 * 256 >= 2^-22 |t| of a half-integer
        if (__builtin_amdgcn_ballot_w64(amb) != 0) {
#pragma unroll
            for (int c = 0; c < 4; ++c)
#pragma unroll
                for (int e = 0; e < 4; ++e) rq[4 * c + e] = __builtin_rintf(v[4 * i4 + c][e] / q_scale) + q_zpf;
        }
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            const int i = 4 * i4 + c;
            int row = (i * 64) / 96, pc = (i * 64) % 96 + lane;          // piece pc (4 values -> 4 bytes) of row `row` of the half
            if (pc >= 96) { pc -= 96; row += 1; }
            uint32_t pk = 0;
#pragma unroll
            for (int e = 0; e < 4; ++e) pk = pack_u8(pk, rq[4 * c + e], e);           // saturating: the operator's clamp to [0, 255]
            *reinterpret_cast<uint32_t *>(dst + row * 384 + (((pc >> 2) ^ (row & 7)) << 4) + ((pc & 3) << 2)) = pk ^ 0x80808080u;
        }
    }
}

// ---- q | k | v + attention, one workgroup per SEQUENCE (all heads) ---------------------------------------------------------------------
// qkv_attn_i8_kernel above runs one workgroup per (sequence, head): every head fetches the sequence's quantised rows again (32 bytes per row and load
// instruction: 32 cache lines per kilobyte) and its own weight fragments once per task; measured 1.8 ms per layer at 4096 padded texts, of which
// ~0.7 ms is the load path. Here a workgroup of EIGHT waves keeps the sequence: wave w holds token block w as twelve int8 MFMA fragments in
// registers for all heads -- quantised on the way in from the f32 layer input (DynamicQuantizeLinear with the tensor's range keys: the separate
// quantising pass over X and its 0.4 GB of bytes are gone) -- and the head's three weight blocks (36 KiB, fragment-major) arrive in LDS by LDS-DMA,
// double-buffered, shared by the eight waves. Per head: every wave projects Q for its block; waves 0-3 project K for key block w, waves 4-7 project
// V^T for key block w - 4 (they keep that block's fragments too), both only for real keys; ONE barrier; then the attention of qkv_attn_i8_kernel
// (same operand layouts, f16 hi/lo split products, f32 softmax). K / V fragments are double-buffered, so a head needs no second barrier.
// Needs: hidden 384, 32-wide heads, <= 256 positions and <= 128 keys per sequence (the tokenizer's window, minilm.rs:112-117); other shapes take
// the per-head kernel.
constexpr int QS_WB = 36 * 1024;                          // one head's q | k | v weight fragments
constexpr int QS_KV = 4 * 8192;                           // K / V fragments of up to 4 key blocks
constexpr int QS_OFF_KV = 2 * QS_WB, QS_OFF_SCR = QS_OFF_KV + 2 * QS_KV, QS_OFF_C = QS_OFF_SCR + 8 * 2048, QS_CB = 4 * 128 * 4, QS_LDS = QS_OFF_C + 2 * QS_CB;
// per-head constants of the fused q|k|v matrix as ONE 2 KiB block [scale | rowsum_w - K z | bias | z][128] (96 used: q, k, v features of the head), so that
// they travel with the weights by LDS-DMA
__global__ void pack_head_consts_kernel(const float *__restrict__ wscale, const int32_t *__restrict__ rsz, const float *__restrict__ bias, const int32_t *__restrict__ zw,
                                        uint32_t *__restrict__ out, int heads, int H) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= heads * 512) return;
    const int h = i >> 9, arr = (i >> 7) & 3, j = i & 127;
    uint32_t v = 0;
    if (j < 96) {
        const int n = (j >> 5) * H + h * 32 + (j & 31);
        if (arr == 0) v = __float_as_uint(wscale[n]); else if (arr == 1) v = (uint32_t)rsz[n]; else if (arr == 2) v = __float_as_uint(bias[n]); else v = zw ? (uint32_t)zw[n] : 0u;
    }
    out[i] = v;
}
#ifdef SHODH_PROF
#define QPROF_DECL long long pt_[10] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0}, tq_ = clock64(); const long long t00_ = tq_, w00_ = wall_clock64();
#define QPROF_T(i) { const long long t_ = clock64(); pt_[i] += t_ - tq_; tq_ = t_; }
#define QPROF_END if (lane == 0 && (blockIdx.x == 7 || blockIdx.x == 2049) ) printf("seqprof blk %d wave %d P %d S %d wall100MHz %lld total %lld : load %lld aux %lld w0 %lld | Q %lld KV %lld vmwait %lld bar %lld issue %lld att %lld out %lld\n", (int)blockIdx.x, wave, P, S, wall_clock64() - w00_, clock64() - t00_, pt_[0], pt_[1], pt_[2], pt_[3], pt_[4], pt_[5], pt_[6], pt_[7], pt_[8], pt_[9]);
#else
#define QPROF_DECL
#define QPROF_T(i)
#define QPROF_END
#endif
template <bool ZW>
__global__ __launch_bounds__(512, 2) void qkv_attn_seq_kernel(const float *__restrict__ X /* f32 layer input [M][384] */, const uint32_t *__restrict__ mmA /* its range keys */,
                                                              const int8_t *__restrict__ Wp /* fused q|k|v weight, fragment-major */, const uint32_t *__restrict__ hconsts /* pack_head_consts_kernel: [heads][4][128] */,
                                                              const int32_t *__restrict__ cu, const int32_t *__restrict__ klen /* or null */, float *__restrict__ ctx,
                                                              uint32_t *__restrict__ mm_out, int heads_all, int mm_stride /* 0: one range for the batch tensor; 2: one per sequence */,
                                                              int head_splits /* workgroups per sequence: each takes heads_all / head_splits heads (a few texts: one sequence's twelve heads one after the other are 50 us per layer) */) {
    constexpr int H = 384, KS = 12;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const uint32_t smem_lds = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) unsigned char *)smem;
    const int seq = blockIdx.x / head_splits;
    const int heads = heads_all / head_splits, h0 = (int)(blockIdx.x % head_splits) * heads;      // this workgroup's heads: h0 .. h0 + heads - 1 (local index h below)
    const int t0 = cu[seq], P = cu[seq + 1] - t0;
    const int S = klen ? klen[seq] : P;
    const int tid = threadIdx.x, lane = tid & 63, hi = lane >> 5, l31 = lane & 31;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int nkb = (S + 31) >> 5, nqb = (P + 31) >> 5;               // nkb <= 4, nqb <= 8 (launcher)
    mmA += (size_t)seq * mm_stride;
    if (mm_out) mm_out += (size_t)seq * mm_stride;
    const ActQ ap = act_params(mmA);
    const float a_scale = ap.scale, zpf = (float)ap.zp;
    const int corr = 128 - ap.zp;
    float omin = __builtin_inff(), omax = -__builtin_inff();      // range of this wave's outputs (order keys only at the very end)
    QPROF_DECL

    // ---- per-head constants and weights into LDS (double-buffered); heads 0 and 1 are requested before the rows are fetched
    const unsigned char *wg = reinterpret_cast<const unsigned char *>(Wp);
    const unsigned char *cg = reinterpret_cast<const unsigned char *>(hconsts);
    auto issue_head = [&](int h, int buf) {                // 38 pieces of 1 KiB: 36 weight fragments (piece i = which * 12 + ks) + the head's 2 KiB of constants; wave w issues pieces w, w + 8, ...
#pragma unroll
        for (int i0 = 0; i0 < 5; ++i0) {
            const int i = i0 * 8 + wave;
            if (i < 36) {
                const int which = i / 12, ks = i % 12;
                const unsigned char *src = uniform_ptr(wg + ((size_t)(which * 12 + h0 + h) * KS + ks) * 1024);
                glds16(src, (uint32_t)lane * 16u, (uint32_t)__builtin_amdgcn_readfirstlane((int)(smem_lds + (uint32_t)buf * QS_WB + (uint32_t)i * 1024u)));
            } else if (i < 38) {
                const unsigned char *src = uniform_ptr(cg + (size_t)(h0 + h) * QS_CB + (size_t)(i - 36) * 1024);
                glds16(src, (uint32_t)lane * 16u, (uint32_t)__builtin_amdgcn_readfirstlane((int)(smem_lds + (uint32_t)QS_OFF_C + (uint32_t)buf * QS_CB + (uint32_t)(i - 36) * 1024u)));
            }
        }
    };
    // Two buffers, two heads ahead: the weights of head h are needed by its projections only, so their buffer is free as soon as every wave has
    // passed the head's barrier -- head h + 2 is requested right there and has a whole head (attention of h, projections of h + 1) to arrive.
    // (Requested at the top of head h + 1 instead, it had only that head's projections, and every head stalled on it: 1.71 ms per layer.)
    issue_head(0, 0);
    if (heads > 1) issue_head(1, 1);

    i32x4q xf_own[KS], xf_aux[KS];
    int rs_own = 0, rs_aux = 0;
    unsigned char *xch = smem + QS_OFF_KV + QS_KV;                               // staged blocks 0-3, 12 KiB each: the second K / V buffer and the scratch behind it (free until the first head's barrier)
    // half a token block (16 rows = 24 KiB of f32, contiguous) of `src` (row pitch 384 floats) as 24 fully coalesced 1-KiB requests per wave
    auto fetch = [&](const float *src, f32x4q (&v)[24], int hf) {
        const int row0 = wave * 32 + hf * 16;
        if (row0 + 16 <= P) fetch_rows16(src + (size_t)(t0 + row0) * H, lane, v);      // wave-uniform: the half block is one contiguous span
        else {                                           // rows past the sequence's end repeat its last row (their results are never used)
#pragma unroll
            for (int i = 0; i < 24; ++i) {
                int row = (i * 64) / 96, pc = (i * 64) % 96 + lane;
                if (pc >= 96) { pc -= 96; row += 1; }
                int tok = row0 + row; if (tok >= P) tok = P - 1; if (tok < 0) tok = 0;
                v[i] = *(reinterpret_cast<const f32x4q *>(src + (size_t)(t0 + tok) * H) + pc);
            }
        }
    };
    // ---- this wave's token block as int8 MFMA fragments: lane (token l31 of the block, half hi), step ks: bytes 32 ks + 16 hi .. + 16 of the row.
    // A workgroup is alone on its CU (LDS) and its eight waves are all it has to cover the latency of HBM, so the block is fetched the way memory likes
    // it and reshaped on chip: a half block arrives fully coalesced, all 48 requests of a block in flight at once; every lane quantises the 16-byte
    // pieces it happens to hold (DynamicQuantizeLinear is elementwise) and drops the four bytes into a row-major staging block in LDS, from which
    // the lanes take their fragments (conflict-free with the swizzle). (s_memtime, per sequence of 180 000 cycles: fetching with lane = row -- 64
    // separate 16-byte requests per instruction -- took 31 000 cycles for one block and as much again for the second block waves 4-7 then needed;
    // 33 000 with the requests batched; this form ...) Waves 0-3 keep their staged block: it is where the V^T tasks below find another wave's rows.
    if (wave < nqb) {
        unsigned char *stg = wave < 4 ? xch + wave * 12288 : smem + QS_OFF_KV + (wave - 4) * 6144;      // waves 4-7: 6 KiB of the first K / V buffer, used for both halves
        const int half_step = wave < 4 ? 6144 : 0;
        f32x4q va[24], vb[24];
        auto stage = [&](const f32x4q (&v)[24], int hf) {
            unsigned char *dst = stg + hf * half_step;
            quant_half16(v, dst, a_scale, zpf, lane);
            if ((l31 >> 4) == hf) {                            // the lanes whose token lies in this half take their fragments
                const unsigned char *src = dst + (l31 & 15) * 384;
#pragma unroll
                for (int ks = 0; ks < KS; ++ks) xf_own[ks] = *reinterpret_cast<const i32x4q *>(src + (((2 * ks + hi) ^ (l31 & 7)) << 4));
            }
        };
        fetch(X, va, 0);
        fetch(X, vb, 1);
        __builtin_amdgcn_sched_barrier(0);
        stage(va, 0);
        stage(vb, 1);
        if (ZW) {                                              // row sums of the stored bytes a - 128
#pragma unroll
            for (int ks = 0; ks < KS; ++ks)
#pragma unroll
                for (int c = 0; c < 4; ++c) rs_own = __builtin_amdgcn_sdot4(xf_own[ks][c], 0x01010101, rs_own, false);
            rs_own += __shfl_xor(rs_own, 32);
        }
    }
    QPROF_T(0)
    // K / V^T tasks of a head: task t < nkb = K of key block t, task nkb + kb = V^T of key block kb; wave t takes task t, so that up to four tasks
    // land on four different SIMDs (waves w and w + 4 share one). K of block w uses the wave's own fragments; V^T of block w - nkb reads that
    // block's staged rows.
    const bool does_k = wave < nkb, does_v = wave >= nkb && wave < 2 * nkb;      // wave-uniform
    const int vkb = wave - nkb;
    QPROF_T(1)
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (does_v) {
        const unsigned char *src = xch + (size_t)vkb * 12288 + l31 * 384;
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
            xf_aux[ks] = *reinterpret_cast<const i32x4q *>(src + (((2 * ks + hi) ^ (l31 & 7)) << 4));
            if (ZW) {
#pragma unroll
                for (int c = 0; c < 4; ++c) rs_aux = __builtin_amdgcn_sdot4(xf_aux[ks][c], 0x01010101, rs_aux, false);
            }
        }
        if (ZW) rs_aux += __shfl_xor(rs_aux, 32);
    }
    QPROF_T(2)

    const float sc = 0.17677669529663688110f * 1.44269504088896340736f;      // 1/sqrt(32) * log2(e): softmax in base 2
    // float(acc + (128 - a_zp) (rowsum_w - K z) - z rowsum_a) * (a_scale * w_scale) + bias for the sixteen features f(r, hi) of a 32-feature block starting at
    // constant slot nb0, two at a time on the packed-FP32 pipe (the operations and their order are those of the scalar form)
    auto dequant_rows = [&](const i32x16l &acc, int nb0, const float *cs, const int32_t *ci, f32x2q (&v2)[8]) {
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const int nl = nb0 + 8 * g + 4 * hi;
            const f32x4q ws = *reinterpret_cast<const f32x4q *>(cs + nl), b4 = *reinterpret_cast<const f32x4q *>(cs + 256 + nl);
            const i32x4q rz = *reinterpret_cast<const i32x4q *>(ci + 128 + nl), z4 = *reinterpret_cast<const i32x4q *>(ci + 384 + nl);
#pragma unroll
            for (int e2 = 0; e2 < 2; ++e2) {
                f32x2q f, w, b;
#pragma unroll
                for (int e = 0; e < 2; ++e) {
                    f[e] = (float)(acc[4 * g + 2 * e2 + e] + __mul24(corr, rz[2 * e2 + e]) - (ZW ? z4[2 * e2 + e] * rs_own : 0));
                    w[e] = ws[2 * e2 + e]; b[e] = b4[2 * e2 + e];
                }
                v2[2 * g + e2] = f * (w * a_scale) + b;
            }
        }
    };
    for (int h = 0; h < heads; ++h) {
        const unsigned char *wb = smem + (h & 1) * QS_WB;
        const float *cs = reinterpret_cast<const float *>(smem + QS_OFF_C + (h & 1) * QS_CB);
        const int32_t *ci = reinterpret_cast<const int32_t *>(cs);
        unsigned char *kv = smem + QS_OFF_KV + (h & 1) * QS_KV;
        // ---- Q for this wave's token block: lane = query, registers = features f(r, hi)
        f16x8q bqh[2], bql[2];
        if (wave < nqb) {
            i32x16l acc;
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[r] = 0;
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) acc = __builtin_amdgcn_mfma_i32_32x32x32_i8(*reinterpret_cast<const i32x4q *>(wb + ks * 1024 + lane * 16), xf_own[ks], acc, 0, 0, 0);
            f32x2q v2[8];
            dequant_rows(acc, 0, cs, ci, v2);
#pragma unroll
            for (int j = 0; j < 8; ++j) v2[j] = v2[j] * sc;
            f16_split8p(v2, bqh[0], bql[0]);
            f16_split8p(v2 + 4, bqh[1], bql[1]);
        }
        QPROF_T(3)
        // ---- K (task waves 0 .. nkb-1: key block w) / V^T (task waves nkb .. 2 nkb-1: key block w - nkb): fragments -> LDS
        if (does_k) {
            i32x16l acc;
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[r] = 0;
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) acc = __builtin_amdgcn_mfma_i32_32x32x32_i8(*reinterpret_cast<const i32x4q *>(wb + (12 + ks) * 1024 + lane * 16), xf_own[ks], acc, 0, 0, 0);
            f32x2q v2[8];
            dequant_rows(acc, 32, cs, ci, v2);
#pragma unroll
            for (int s2 = 0; s2 < 2; ++s2) {
                f16x8q fh, fl;
                f16_split8p(v2 + 4 * s2, fh, fl);
                unsigned char *dst = kv + (size_t)((wave * 2 + 0) * 2 + s2) * 2048 + lane * 16;
                *reinterpret_cast<f16x8q *>(dst) = fh;
                *reinterpret_cast<f16x8q *>(dst + 1024) = fl;
            }
        }
        if (does_v) {                                     // operands swapped: lane = feature d, registers = keys f(r, hi) of the block
            const int kb = vkb;
            i32x16l acc;
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[r] = 0;
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) acc = __builtin_amdgcn_mfma_i32_32x32x32_i8(xf_aux[ks], *reinterpret_cast<const i32x4q *>(wb + (24 + ks) * 1024 + lane * 16), acc, 0, 0, 0);
            const float ws1 = a_scale * cs[64 + l31], b1 = cs[256 + 64 + l31];
            const int rz1 = __mul24(corr, ci[128 + 64 + l31]), z1 = ci[384 + 64 + l31];
            f32x2q v2[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                f32x2q f;
#pragma unroll
                for (int e = 0; e < 2; ++e) {
                    const int r = 2 * j + e;
                    const int tl = (r & 3) + 8 * (r >> 2) + 4 * hi;       // token of the block whose value register r holds
                    const int rsa = ZW ? __shfl(rs_aux, tl) : 0;           // (lane tl of this wave holds that token's row sum)
                    f[e] = (float)(acc[r] + rz1 - (ZW ? z1 * rsa : 0));
                }
                v2[j] = f * ws1 + b1;
            }
#pragma unroll
            for (int s2 = 0; s2 < 2; ++s2) {
                f16x8q fh, fl;
                f16_split8p(v2 + 4 * s2, fh, fl);
                unsigned char *dst = kv + (size_t)((kb * 2 + 1) * 2 + s2) * 2048 + lane * 16;
                *reinterpret_cast<f16x8q *>(dst) = fh;
                *reinterpret_cast<f16x8q *>(dst + 1024) = fl;
            }
        }
        QPROF_T(4)
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // this wave's pieces of the next head (requested a head ago) and its stores
        QPROF_T(5)
        __syncthreads();
        QPROF_T(6)
        if (h + 2 < heads) issue_head(h + 2, h & 1);
        QPROF_T(7)
        // ---- attention of this wave's query block over the real keys. A sequence has at most four key blocks, so the softmax is taken in one piece:
        // all score blocks first (independent accumulators, their MFMAs interleaved so that no product waits for the previous one), one maximum, one
        // pass of exponentials, then P V on two accumulators. (Block after block with a running maximum, every product chain was dependent and every
        // block paid a rescale of the output: 1 700 cycles per key block against ~1 200 here.)
        if (wave < nqb) {
            const int q = wave * 32 + l31;
            f32x16q o, o2;
#pragma unroll
            for (int r = 0; r < 16; ++r) { o[r] = 0.0f; o2[r] = 0.0f; }
            float l = 0.0f;
            auto attend = [&](auto nk_c) {
                constexpr int NK = decltype(nk_c)::value;
                f32x16q st[NK];
#pragma unroll
                for (int kb = 0; kb < NK; ++kb)
#pragma unroll
                    for (int r = 0; r < 16; ++r) st[kb][r] = 0.0f;
#pragma unroll
                for (int s2 = 0; s2 < 2; ++s2) {
                    f16x8q kh[NK], kl[NK];
#pragma unroll
                    for (int kb = 0; kb < NK; ++kb) {
                        const unsigned char *kf = kv + (size_t)(kb * 2) * 4096 + lane * 16 + s2 * 2048;
                        kh[kb] = *reinterpret_cast<const f16x8q *>(kf); kl[kb] = *reinterpret_cast<const f16x8q *>(kf + 1024);
                    }
#pragma unroll
                    for (int kb = 0; kb < NK; ++kb) st[kb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(kh[kb], bqh[s2], st[kb], 0, 0, 0);
#pragma unroll
                    for (int kb = 0; kb < NK; ++kb) st[kb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(kh[kb], bql[s2], st[kb], 0, 0, 0);
#pragma unroll
                    for (int kb = 0; kb < NK; ++kb) st[kb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(kl[kb], bqh[s2], st[kb], 0, 0, 0);
                }
#pragma unroll
                for (int r = 0; r < 16; ++r) {                  // only the last key block can hold masked keys
                    const int key = (NK - 1) * 32 + (r & 3) + 8 * (r >> 2) + 4 * hi;
                    st[NK - 1][r] = key < S ? st[NK - 1][r] : -3.0e38f;
                }
                float mx = -3.0e38f;
#pragma unroll
                for (int kb = 0; kb < NK; ++kb)
#pragma unroll
                    for (int r = 0; r < 16; ++r) mx = fmaxf(mx, st[kb][r]);
                mx = fmaxf(mx, __shfl_xor(mx, 32));
                f32x2q ps2 = 0.0f;                             // (subtraction and row sum two values at a time; v_exp_f32 has no packed form)
#pragma unroll
                for (int kb = 0; kb < NK; ++kb)
#pragma unroll
                    for (int j = 0; j < 8; ++j) {
                        f32x2q d2; d2[0] = st[kb][2 * j]; d2[1] = st[kb][2 * j + 1];
                        d2 = d2 - mx;
                        f32x2q e2; e2[0] = __builtin_amdgcn_exp2f(d2[0]); e2[1] = __builtin_amdgcn_exp2f(d2[1]);
                        ps2 = ps2 + e2;
                        st[kb][2 * j] = e2[0]; st[kb][2 * j + 1] = e2[1];
                    }
                const float ps = ps2[0] + ps2[1];
                l = ps + __shfl_xor(ps, 32);
#pragma unroll
                for (int kb = 0; kb < NK; ++kb) {
                    const unsigned char *vf = kv + (size_t)(kb * 2 + 1) * 4096 + lane * 16;
                    f32x2q p2[8];
#pragma unroll
                    for (int j = 0; j < 8; ++j) { p2[j][0] = st[kb][2 * j]; p2[j][1] = st[kb][2 * j + 1]; }
                    f16x8q ph[2], pl[2], vh[2], vl[2];
#pragma unroll
                    for (int s2 = 0; s2 < 2; ++s2) {
                        f16_split8p(p2 + 4 * s2, ph[s2], pl[s2]);
                        vh[s2] = *reinterpret_cast<const f16x8q *>(vf + s2 * 2048); vl[s2] = *reinterpret_cast<const f16x8q *>(vf + s2 * 2048 + 1024);
                    }
                    o = __builtin_amdgcn_mfma_f32_32x32x16_f16(vh[0], ph[0], o, 0, 0, 0);
                    o2 = __builtin_amdgcn_mfma_f32_32x32x16_f16(vh[1], ph[1], o2, 0, 0, 0);
                    o = __builtin_amdgcn_mfma_f32_32x32x16_f16(vh[0], pl[0], o, 0, 0, 0);
                    o2 = __builtin_amdgcn_mfma_f32_32x32x16_f16(vh[1], pl[1], o2, 0, 0, 0);
                    o = __builtin_amdgcn_mfma_f32_32x32x16_f16(vl[0], ph[0], o, 0, 0, 0);
                    o2 = __builtin_amdgcn_mfma_f32_32x32x16_f16(vl[1], ph[1], o2, 0, 0, 0);
                }
            };
            switch (nkb) {                                    // wave-uniform
                case 1: attend(std::integral_constant<int, 1>{}); break;
                case 2: attend(std::integral_constant<int, 2>{}); break;
                case 3: attend(std::integral_constant<int, 3>{}); break;
                case 4: attend(std::integral_constant<int, 4>{}); break;
                default: break;                               // no keys: l = 0, the row below becomes 0 * inf as before
            }
#pragma unroll
            for (int r = 0; r < 16; ++r) o[r] += o2[r];
            QPROF_T(8)
            const float invl = 1.0f / l;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                o[r] *= invl;
                if (q < P) { omin = fminf(omin, o[r]); omax = fmaxf(omax, o[r]); }
            }
            // O^T: this lane = query q, register r = feature f(r, hi): four runs of four consecutive features, stored as they are -- the four stores
            // of a wave complete every row's 128-byte line of this head. (Through a wave-private LDS transposition, so that each store instruction
            // wrote whole lines, the epilogue took 1 450 cycles per head instead of 600.)
            if (q < P) {
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    f32x4q ov;
#pragma unroll
                    for (int e = 0; e < 4; ++e) ov[e] = o[4 * g + e];
                    *reinterpret_cast<f32x4q *>(ctx + (size_t)(t0 + q) * H + (h0 + h) * 32 + 8 * g + 4 * hi) = ov;
                }
            }
        }
        QPROF_T(9)
    }
    QPROF_END
    if (mm_out) {
        uint32_t klo = omin <= omax ? order_key(omin) : 0xFFFFFFFFu, khi = omin <= omax ? order_key(omax) : 0u;
        for (int ofs = 32; ofs > 0; ofs >>= 1) { klo = min(klo, (uint32_t)__shfl_xor((int)klo, ofs)); khi = max(khi, (uint32_t)__shfl_xor((int)khi, ofs)); }
        if (lane == 0) {
            if (klo < __atomic_load_n(mm_out, __ATOMIC_RELAXED)) atomicMin(mm_out, klo);
            if (khi > __atomic_load_n(mm_out + 1, __ATOMIC_RELAXED)) atomicMax(mm_out + 1, khi);
        }
    }
}

// ---- attention output projection + residual + LayerNorm + DynamicQuantizeLinear of the result, one workgroup per SEQUENCE (round 4) -----------------
// SHODH_QUANT_SCOPE_PER_TEXT only: every range of the layer's first half is local to one sequence, so what the batch scope runs as four launches
// (act_quant over CTX, i8_stream_kernel<RESID_LN>, act_quant over the LayerNorm output, and the range atomics that tie them together: 0.43 + 0.78 +
// 0.43 ms per layer at 4096 texts, 7.5 GB of traffic) is one kernel without a global atomic:
//   the sequence's ctx range (left by qkv_attn_seq_kernel) -> the eight waves fetch the sequence's ctx rows the way the layer input is fetched
//   (fetch_rows16 / quant_half16: coalesced 1-KiB requests, quantised in the lane that holds them, row-major swizzled staging blocks in LDS)
//   -> out^T = W_o . ctx^T with k OUTERMOST. A PAIR of waves (sharing a SIMD pair) owns a 32-token block: wave fh of the pair keeps the 6 x 16
//   accumulators of feature blocks 6 fh .. 6 fh + 5 (96 registers), per k-step one token fragment from the block's staging bytes and six weight
//   fragments from a 4-slot ring of 12-KiB k-slices that LDS-DMA keeps three slices ahead (W_o, 144 KiB, cannot sit in LDS next to 96 KiB of ctx bytes)
//   -> dequantise + bias + residual (all 24 residual loads of the wave's half rows in flight at once) -> LayerNorm: a token's row lies in the two
//   lane pairs of the wave pair; the block sums are chained in block order THROUGH the pair (first half, hand-over in LDS, second half: i8_stream_kernel's
//   order of additions, the same bits) -> f32 rows out through a 4-KiB LDS tile per wave (whole 128-byte segments per store, in place over X)
//   -> the sequence's output range over the eight waves (LDS) -> the waves fetch the rows back (their L2 holds them) through fetch_rows16 / quant_half16
//   with the new range -> whole rows of XQ, row sums, and the range keys for the FFN kernels.
// Tried first: four waves with a whole row per lane pair (192 accumulators, a lone wave's 512 registers): the compiler kept the accumulators in AGPRs,
// copied them out for the LayerNorm and spilled around them, and every spill reload is a vmcnt(0) in front of the residual loads: 245 000 cycles per
// sequence (2.0 ms per layer -- no faster than the four launches it replaces). And inside qkv_attn_seq_kernel (256 registers) the same code spilled 2.3 KB per lane: 4.8 ms.
#ifdef SHODH_PROF
#define OPROF_DECL long long pt_[10] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0}, tq_ = clock64(); const long long t00_ = tq_;
#define OPROF_T(i) { const long long t_ = clock64(); pt_[i] += t_ - tq_; tq_ = t_; }
#define OPROF_END if (lane0 == 0 && (blockIdx.x == 7 || blockIdx.x == 2049)) printf("outprof blk %d wave %d total %lld : ctx stage %lld | wait+bar %lld gemm %lld dequant+resid %lld mean chain %lld var chain %lld norm+store %lld | range bar %lld refetch+quant+out %lld\n", (int)blockIdx.x, wave, clock64() - t00_, pt_[0], pt_[1], pt_[2], pt_[3], pt_[4], pt_[5], pt_[6], pt_[7], pt_[8]);
#else
#define OPROF_DECL
#define OPROF_T(i)
#define OPROF_END
#endif
struct AttnOutArgs {
    const float *ctx; const uint32_t *mm_ctx;     // attention output [M][384], its range keys [sequences][2]
    const int8_t *Wp;                             // W_o fragment-major [12][12][64][16] (pack_i8_frag_kernel)
    const float *wscale; const int32_t *rsz; const int32_t *zw /* or null */; const float *bias, *gamma, *beta; float eps;
    float *X;                                     // residual in, LayerNorm output out (in place)
    int8_t *XQ; int32_t *rsq /* or null */;       // DynamicQuantizeLinear(LayerNorm output) [M][384] (signed storage), its row sums
    uint32_t *mm_x1;                              // [sequences][2] range keys of the LayerNorm output
    int rows;                                     // positions per sequence: 128 or 256
};
constexpr int OT_SLOT = 12 * 1024, OT_NSLOT = 4, OT_STG = OT_NSLOT * OT_SLOT, OT_CONST = OT_STG + 8 * 12288, OT_RED = OT_CONST + 6 * 1536, OT_LDS = OT_RED + 64 + 2 * 4 * 32 * 4;      // 154 KiB
template <int BPW /* token blocks per wave pair: 2 (256 positions) or 1 (128) */>
__global__ __launch_bounds__(512, 2) void attn_out_ln_quant_seq_kernel(const AttnOutArgs a) {
    constexpr int H = 384, KS = 12, NBW = 6, NBLK = 4 * BPW;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const uint32_t smem_lds = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) unsigned char *)smem;
    const int seq = blockIdx.x;
    const int tid = threadIdx.x, lane0 = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int pr = wave >> 1, fh = wave & 1;                             // wave pair, feature half
    const int s0 = seq * a.rows;                                         // the sequence's first row
    // The lane id is re-read through an empty asm at the top of every stage: otherwise the compiler computes every swizzled LDS address of the kernel
    // (hundreds of loop-invariant values) up front and parks them in scratch next to the accumulators
    int lane = lane0, hi = lane0 >> 5, l31 = lane0 & 31;
#define OT_FRESH_LANE() { lane = lane0; asm volatile("" : "+v"(lane)); hi = lane >> 5; l31 = lane & 31; }
    float *t_red = reinterpret_cast<float *>(smem + OT_RED);            // [8 waves][min, max]
    float *ex_a = t_red + 16, *ex_m = ex_a + 4 * 32;                      // [4 pairs][32 tokens]: the chain's hand-over, the finished statistic
    float *c_ws = reinterpret_cast<float *>(smem + OT_CONST);
    int32_t *c_rz = reinterpret_cast<int32_t *>(c_ws + H);
    float *c_b = reinterpret_cast<float *>(c_rz + H);
    int32_t *c_zw = reinterpret_cast<int32_t *>(c_b + H);
    float *c_g = reinterpret_cast<float *>(c_zw + H), *c_be = c_g + H;
    const bool zwo = a.zw != nullptr;
    // k-slice ks of W_o (twelve 1-KiB fragment blocks, one per 32 output features) -> ring slot ks & 3: waves 0-3 request blocks w and w + 8, waves 4-7 block w
    auto issue_wslice = [&](int ks) {
        const unsigned char *wo = reinterpret_cast<const unsigned char *>(a.Wp);
        const uint32_t slot = smem_lds + (uint32_t)((ks & (OT_NSLOT - 1)) * OT_SLOT);
        glds16(uniform_ptr(wo + ((size_t)wave * KS + ks) * 1024), (uint32_t)lane0 * 16u, (uint32_t)__builtin_amdgcn_readfirstlane((int)(slot + (uint32_t)wave * 1024u)));
        if (wave < 4) glds16(uniform_ptr(wo + ((size_t)(wave + 8) * KS + ks) * 1024), (uint32_t)lane0 * 16u, (uint32_t)__builtin_amdgcn_readfirstlane((int)(slot + (uint32_t)(wave + 8) * 1024u)));
    };
    OPROF_DECL
    issue_wslice(0); issue_wslice(1); issue_wslice(2);
    const ActQ cp = act_params(a.mm_ctx + 2 * seq);
    const float c_scale = cp.scale;
    const int c_corr = 128 - cp.zp;
    for (int i = tid; i < H; i += 512) {      // per-feature constants with the sequence's activation parameters folded in, as i8_stream_kernel folds the tensor's
        c_ws[i] = c_scale * a.wscale[i]; c_rz[i] = c_corr * a.rsz[i]; c_b[i] = a.bias[i]; c_zw[i] = zwo ? a.zw[i] : 0; c_g[i] = a.gamma[i]; c_be[i] = a.beta[i];
    }
    // ---- the sequence's ctx rows -> bytes in the staging blocks: 2 NBLK half blocks over eight waves
    {
        OT_FRESH_LANE();
        f32x4q v0[24], v1[24];                                           // (all of the wave's requests in flight before the first byte is converted)
        fetch_rows16(a.ctx + (size_t)(s0 + wave * 16) * H, lane, v0);
        if (BPW == 2) fetch_rows16(a.ctx + (size_t)(s0 + (wave + 8) * 16) * H, lane, v1);
        __builtin_amdgcn_sched_barrier(0);
        quant_half16(v0, smem + OT_STG + (wave >> 1) * 12288 + (wave & 1) * 6144, c_scale, (float)cp.zp, lane);
        if (BPW == 2) { OT_FRESH_LANE(); quant_half16(v1, smem + OT_STG + ((wave + 8) >> 1) * 12288 + (wave & 1) * 6144, c_scale, (float)cp.zp, lane); }
    }
    OPROF_T(0)
    float xmin = __builtin_inff(), xmax = -__builtin_inff();
#pragma unroll 1
    for (int b = 0; b < BPW; ++b) {
        const int tb = pr * BPW + b;                                     // this pair's token block
        // reg[j]: the 32 x 32 block (features 32 (6 fh + j) .., the block's tokens): lane = token, registers = features f(r, hi); int32 accumulators first,
        // the f32 results of the later stages in the same registers
        f32x16q reg[NBW];
#pragma unroll
        for (int j = 0; j < NBW; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) reg[j][r] = 0.0f;               // (+0.0f == int 0)
        int rs_t = 0;
        unsigned char *sblk = smem + OT_STG + tb * 12288;
        const bool more = b + 1 < BPW;                                   // uniform
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                 // slices 0-2 of this pass (and everything older) have landed
        __syncthreads();
        OPROF_T(1)
        static_for_i(std::make_integer_sequence<int, KS>{}, [&](auto ks_c) {
            constexpr int ks = decltype(ks_c)::value;
            if (ks > 0) {
                // slice ks must have landed for every wave, and slot (ks - 1) & 3 must be drained before it is refilled: this wave's pieces of the slices after
                // ks (at most two of them: ks + 1, ks + 2; two pieces each for waves 0-3, one for waves 4-7) may stay in flight
                constexpr int after = KS - 1 - ks < 2 ? KS - 1 - ks : 2;
                if (wave < 4) asm volatile("s_waitcnt vmcnt(%0)" ::"i"(2 * after) : "memory"); else asm volatile("s_waitcnt vmcnt(%0)" ::"i"(after) : "memory");
                __builtin_amdgcn_s_waitcnt(0xC07F);
                __builtin_amdgcn_s_barrier();
                __builtin_amdgcn_sched_barrier(0);
            }
            if (ks + 3 < KS) issue_wslice(ks + 3);
            OT_FRESH_LANE();
            const i32x4q xf = *reinterpret_cast<const i32x4q *>(sblk + l31 * 384 + (((2 * ks + hi) ^ (l31 & 7)) << 4));
            if (zwo) {
#pragma unroll
                for (int c = 0; c < 4; ++c) rs_t = __builtin_amdgcn_sdot4(xf[c], 0x01010101, rs_t, false);
            }
            const unsigned char *ring = smem + (ks & (OT_NSLOT - 1)) * OT_SLOT + fh * (NBW * 1024) + lane * 16;
#pragma unroll
            for (int j = 0; j < NBW; ++j)
                reg[j] = __builtin_bit_cast(f32x16q, __builtin_amdgcn_mfma_i32_32x32x32_i8(*reinterpret_cast<const i32x4q *>(ring + j * 1024), xf, __builtin_bit_cast(i32x16l, reg[j]), 0, 0, 0));
        });
        __syncthreads();                                                 // every wave is through the k loop: the ring and the staging bytes of the current blocks are free
        OPROF_T(2)
        // ---- + bias + residual; block sums. The residual rows (this wave's half: 32 rows x 768 B) come as 24 coalesced requests -- 384-byte row segments,
        // three whole cache lines each -- and are turned into the accumulator layout through 12 KiB of LDS (the wave's share of the ring and of the dead
        // staging bytes; 16-byte chunks XOR-swizzled by row & 7), half of the columns at a time. Read straight in the accumulator layout an instruction
        // touches 32 rows x 32 bytes: 25 000 cycles per block on the address path where this takes a round trip.
        OT_FRESH_LANE();
        const int rsa = zwo ? rs_t + __shfl_xor(rs_t, 32) : 0;
        float sb[NBW];
        {
            f32x4q rr[NBW][4];
            {
                const unsigned char *xb = reinterpret_cast<const unsigned char *>(a.X + (size_t)(s0 + tb * 32) * H) + fh * (NBW * 128);
#pragma unroll
                for (int c = 0; c < 2; ++c)
#pragma unroll
                    for (int i = 0; i < 12; ++i) {
                        const int p = i * 64 + lane, row = p / 24, cc = p % 24;
                        rr[c * 3 + i / 4][i % 4] = *reinterpret_cast<const f32x4q *>(xb + (size_t)row * (H * 4) + c * 384 + cc * 16);
                    }
                unsigned char *lo = smem + wave * 6144, *up = sblk + fh * 6144;      // rows 0-15 | 16-31 of the tile
#pragma unroll
                for (int c = 0; c < 2; ++c) {
#pragma unroll
                    for (int i = 0; i < 12; ++i) {
                        const int p = i * 64 + lane, row = p / 24, cc = p % 24;
                        *reinterpret_cast<f32x4q *>((row < 16 ? lo : up) + (row & 15) * 384 + (((cc & ~7) | ((cc & 7) ^ (row & 7))) << 4)) = rr[c * 3 + i / 4][i % 4];
                    }
                    const unsigned char *rb = (l31 < 16 ? lo : up) + (l31 & 15) * 384;
#pragma unroll
                    for (int jj = 0; jj < 3; ++jj)
#pragma unroll
                        for (int g = 0; g < 4; ++g) rr[c * 3 + jj][g] = *reinterpret_cast<const f32x4q *>(rb + ((jj * 8 + ((2 * g + hi) ^ (l31 & 7))) << 4));
                }
            }
#pragma unroll
            for (int j = 0; j < NBW; ++j) {
                const i32x16l acc = __builtin_bit_cast(i32x16l, reg[j]);
                float s1 = 0.0f;
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const int nl = (fh * NBW + j) * 32 + 8 * g + 4 * hi;
                    const f32x4q ws = *reinterpret_cast<const f32x4q *>(c_ws + nl), b4 = *reinterpret_cast<const f32x4q *>(c_b + nl);
                    const i32x4q rz = *reinterpret_cast<const i32x4q *>(c_rz + nl), z4 = *reinterpret_cast<const i32x4q *>(c_zw + nl);
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const float v = (float)(acc[4 * g + e] + rz[e] - (zwo ? z4[e] * rsa : 0)) * ws[e] + b4[e] + rr[j][g][e];
                        reg[j][4 * g + e] = v; s1 += v;
                    }
                }
                s1 += __shfl_xor(s1, 32);
                sb[j] = s1;
            }
        }
        OPROF_T(3)
        // ---- mean: 0 + s_0 + s_1 + ... + s_11 in block order, first half in wave 0 of the pair, handed over through LDS, second half in wave 1
        float mean;
        {
            float ch = 0.0f;
            if (fh == 0) {
#pragma unroll
                for (int j = 0; j < NBW; ++j) ch += sb[j];
                if (hi == 0) ex_a[pr * 32 + l31] = ch;
            }
            __syncthreads();
            if (more) { issue_wslice(0); issue_wslice(1); issue_wslice(2); }      // (every wave is done with its share of the ring: the next pass's first slices travel under the rest of this epilogue)
            if (fh == 1) {
                ch = ex_a[pr * 32 + l31];
#pragma unroll
                for (int j = 0; j < NBW; ++j) ch += sb[j];
                ch *= (1.0f / (float)H);
                if (hi == 0) ex_m[pr * 32 + l31] = ch;
            }
            __syncthreads();
            mean = ex_m[pr * 32 + l31];
        }
        OPROF_T(4)
        float inv;
        {
            float qb[NBW];
#pragma unroll
            for (int j = 0; j < NBW; ++j) {
                float q1 = 0.0f;
#pragma unroll
                for (int r = 0; r < 16; ++r) { reg[j][r] -= mean; q1 += reg[j][r] * reg[j][r]; }
                q1 += __shfl_xor(q1, 32);
                qb[j] = q1;
            }
            float ch = 0.0f;
            __syncthreads();                                             // (every wave has read the mean: ex_a / ex_m are free again)
            if (fh == 0) {
#pragma unroll
                for (int j = 0; j < NBW; ++j) ch += qb[j];
                if (hi == 0) ex_a[pr * 32 + l31] = ch;
            }
            __syncthreads();
            if (fh == 1) {
                ch = ex_a[pr * 32 + l31];
#pragma unroll
                for (int j = 0; j < NBW; ++j) ch += qb[j];
                if (hi == 0) ex_m[pr * 32 + l31] = ch;
            }
            __syncthreads();
            const float var = ex_m[pr * 32 + l31];
            inv = 1.0f / sqrtf(var * (1.0f / (float)H) + a.eps);
        }
        OPROF_T(5)
        // The rows leave through LDS: in the accumulator layout a store instruction writes 16-byte pieces scattered over 32 rows, which the write path digests
        // at ~4 B/clk per CU. The block's staging bytes are dead after the k loop (the chain's barriers lie in between): each wave of the pair turns its
        // feature blocks through a 4-KiB tile there (32 rows x 128 B, 16-byte chunks XOR-swizzled by row & 7) and every store instruction writes eight whole
        // 128-byte row segments.
        {
            unsigned char *tile = sblk + fh * 6144;
            float *xblk = a.X + (size_t)(s0 + tb * 32) * H + fh * (NBW * 32);
#pragma unroll
            for (int j = 0; j < NBW; ++j) {
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const int nl = (fh * NBW + j) * 32 + 8 * g + 4 * hi;
                    const f32x4q g4 = *reinterpret_cast<const f32x4q *>(c_g + nl), be4 = *reinterpret_cast<const f32x4q *>(c_be + nl);
                    f32x4q ov;
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        ov[e] = reg[j][4 * g + e] * inv * g4[e] + be4[e];
                        xmin = fminf(xmin, ov[e]); xmax = fmaxf(xmax, ov[e]);
                    }
                    *reinterpret_cast<f32x4q *>(tile + l31 * 128 + (((2 * g + hi) ^ (l31 & 7)) << 4)) = ov;
                }
#pragma unroll
                for (int h = 0; h < 4; ++h) {
                    const int tl = h * 8 + (lane >> 3), ch = lane & 7;
                    const f32x4q v4 = *reinterpret_cast<const f32x4q *>(tile + tl * 128 + ((ch ^ (tl & 7)) << 4));
                    *reinterpret_cast<f32x4q *>(xblk + (size_t)tl * H + j * 32 + ch * 4) = v4;
                }
            }
        }
        OPROF_T(6)
    }
    // ---- the sequence's range of the LayerNorm output over the eight waves; then the waves fetch the rows back (L2 holds them) and quantise them on the way
    // through the staging blocks, as the ctx rows were: whole rows of XQ leave as 16-byte pieces
    for (int ofs = 32; ofs > 0; ofs >>= 1) { xmin = fminf(xmin, __shfl_xor(xmin, ofs)); xmax = fmaxf(xmax, __shfl_xor(xmax, ofs)); }
    if (lane0 == 0) { t_red[2 * wave] = xmin; t_red[2 * wave + 1] = xmax; }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                     // this wave's rows are written
    __syncthreads();
    OPROF_T(7)
    float ymin = __builtin_inff(), ymax = -__builtin_inff();
#pragma unroll
    for (int w = 0; w < 8; ++w) { ymin = fminf(ymin, t_red[2 * w]); ymax = fmaxf(ymax, t_red[2 * w + 1]); }
    uint32_t ykeys[2] = {order_key(ymin), order_key(ymax)};
    if (tid == 0) { a.mm_x1[2 * seq] = ykeys[0]; a.mm_x1[2 * seq + 1] = ykeys[1]; }
    const ActQ yp = act_params(ykeys);
    {
        OT_FRESH_LANE();
        f32x4q v0[24], v1[24];
        fetch_rows16(a.X + (size_t)(s0 + wave * 16) * H, lane, v0);
        if (BPW == 2) fetch_rows16(a.X + (size_t)(s0 + (wave + 8) * 16) * H, lane, v1);
        __builtin_amdgcn_sched_barrier(0);
        quant_half16(v0, smem + OT_STG + (wave >> 1) * 12288 + (wave & 1) * 6144, yp.scale, (float)yp.zp, lane);
        if (BPW == 2) { OT_FRESH_LANE(); quant_half16(v1, smem + OT_STG + ((wave + 8) >> 1) * 12288 + (wave & 1) * 6144, yp.scale, (float)yp.zp, lane); }
    }
#pragma unroll 1
    for (int u = wave; u < 2 * NBLK; u += 8) {
        OT_FRESH_LANE();
        unsigned char *dst = smem + OT_STG + (u >> 1) * 12288 + (u & 1) * 6144;
        __builtin_amdgcn_s_waitcnt(0xC07F);
        __builtin_amdgcn_wave_barrier();
        if (a.rsq) {                                                     // row sums of the stored bytes: lane = (row lane & 15, quarter lane >> 4 of its 24 chunks)
            const int r = lane & 15, qd = lane >> 4;
            int rsum = 0;
#pragma unroll
            for (int c6 = 0; c6 < 6; ++c6) {
                const i32x4q x4 = *reinterpret_cast<const i32x4q *>(dst + r * 384 + (((qd * 6 + c6) ^ (r & 7)) << 4));
#pragma unroll
                for (int c = 0; c < 4; ++c) rsum = __builtin_amdgcn_sdot4(x4[c], 0x01010101, rsum, false);
            }
            rsum += __shfl_xor(rsum, 16);
            rsum += __shfl_xor(rsum, 32);
            if (lane < 16) a.rsq[s0 + u * 16 + r] = rsum;
        }
        unsigned char *xq = reinterpret_cast<unsigned char *>(a.XQ) + (size_t)(s0 + u * 16) * H;      // the half block's 16 rows are 6 KiB contiguous
#pragma unroll
        for (int pc = 0; pc < 6; ++pc) {
            const int B = pc * 1024 + lane * 16, row = B / 384, ch = (B % 384) >> 4;
            *reinterpret_cast<u32x4qq *>(xq + B) = *reinterpret_cast<const u32x4qq *>(dst + row * 384 + ((ch ^ (row & 7)) << 4));
        }
    }
    OPROF_T(8)
    OPROF_END
#undef OT_FRESH_LANE
}

// ---- weight-stationary int8 GEMM, K = 384, 384 output features per workgroup ---------------------------------------------------
constexpr int S8_TR = 64, S8_PITCH = 384, S8_TILE = S8_TR * S8_PITCH, S8_NBUF = 3, S8_KS = 12, S8_NT = 768, S8_NPC = 6, S8_NF = 384;
constexpr int S8_CONST = S8_NBUF * S8_TILE;               // wscale | rsz | bias | zw | gamma | beta: 6 x 384 x 4 B
constexpr int S8_RED = S8_CONST + 6 * S8_NF * 4;          // [2][12 waves][64 tokens] f32
constexpr int S8_SCR = S8_RED + 2 * 12 * 64 * 4;          // 12 x 4 KiB wave scratch (LayerNorm epilogue) / 2 x 24 KiB output tiles (quantising epilogue)
constexpr int S8_RS = S8_SCR + 12 * 4096;               // row sums of the three tiles in flight (weight zero points only): [S8_NBUF][256] int32, 64 used
constexpr int S8_LDS = S8_RS + S8_NBUF * 1024;
// The integer terms of the epilogues, (128 - a_zp) * (rowsum_w - K z) and z * rowsum_a, are 24-bit x 24-bit products (|128 - a_zp| <= 128, |z| <= 255,
// |rowsum_w - K z| <= 1536 * 383 < 2^20, |rowsum_a| <= 1536 * 128 < 2^18): v_mul_i32_i24 (__mul24, full rate) gives them exactly where v_mul_lo_u32 runs at a
// quarter of the rate -- with weight zero points every output value carries one or two of them.
enum { SEPI_RESID_LN = 0, SEPI_GELU_RANGE = 1, SEPI_GELU_QUANT = 2 };

struct S8Args {
    const int8_t *XQ; const int32_t *rsA; const uint32_t *mmA;        // quantised activations [M + pad][384], their row sums (or null), their range keys
    const int8_t *Wp; const float *wscale; const int32_t *rsz; const int32_t *zw; const float *bias;      // fragment-major weight + per-feature constants [N]
    const float *resid; const float *gamma; const float *beta; float eps; float *out_f;                    // SEPI_RESID_LN: out_f may alias resid (in place)
    const uint32_t *mmO; int8_t *out_q; int32_t *rs_out;              // SEPI_GELU_QUANT: the output's range (from the RANGE pass), bytes [M][N], row sums (atomicAdd; or null)
    uint32_t *mm_out;                                                  // SEPI_RESID_LN / SEPI_GELU_RANGE: range keys of the output
    int M, N, n_groups;
    int zw_float;                                                      // i8_stream_gelu_kernel: the zero-point terms in float arithmetic (exact for this tensor: QWeight::zw_bound < 2^24)
    int mm_rows;                                                       // PS kernels: rows per sequence (a multiple of the 64-row tile); range slot of row m = m / mm_rows: mmA / mmO / mm_out are [sequences][2], the GELU stats [sequences][4]
};
// range slot (in uint32 words, pairs) of the tile that starts at row m
__device__ __forceinline__ int ps_slot(int m, int mm_rows) { return __builtin_amdgcn_readfirstlane(m / mm_rows); }

// ZW: some weight zero point is non-zero (the row-sum term exists); without it the integer multiply per value is compiled out
// PS: one range per sequence (SHODH_QUANT_SCOPE_PER_TEXT): the activation scale / zero point are per TILE (a.mm_rows is a multiple of the tile), so
// they are applied in the epilogue instead of being folded into the per-feature constants, and the output range is committed per tile
template <int EPI, bool ZW, bool PS = false>
__global__ __launch_bounds__(768, 3) void i8_stream_kernel(const S8Args a) {
    constexpr int KS = S8_KS, NS = 2 * KS, D = 5, RING = 6, PF = S8_NBUF - 1, NPC = S8_NPC;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const uint32_t smem_lds = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) unsigned char *)smem;
    const int tid = threadIdx.x, lane = tid & 63, hi = lane >> 5, l31 = lane & 31;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int M = a.M;
    // XCD-aware block -> (feature group, worker): the n_groups workgroups streaming the same token tiles share an XCD (see gemm_k384_stream_kernel)
    const int n_workers = gridDim.x / a.n_groups;
    const int xcd = blockIdx.x & 7, rr = blockIdx.x >> 3;
    const int grp = rr % a.n_groups, worker = (rr / a.n_groups) * 8 + xcd;
    const int nbase = grp * S8_NF;                        // this workgroup's first feature
    const int nblk = grp * 12 + wave;                     // this wave's 32-feature block

    float *c_ws = reinterpret_cast<float *>(smem + S8_CONST);
    int32_t *c_rz = reinterpret_cast<int32_t *>(c_ws + S8_NF);
    float *c_b = reinterpret_cast<float *>(c_rz + S8_NF);
    int32_t *c_zw = reinterpret_cast<int32_t *>(c_b + S8_NF);
    float *c_g = reinterpret_cast<float *>(c_zw + S8_NF), *c_be = c_g + S8_NF;
    static_assert(!PS || EPI == SEPI_RESID_LN, "the per-sequence form exists for the LayerNorm epilogue (the GELU passes run i8_stream_gelu_kernel)");
    const ActQ ap = PS ? ActQ{1.0f, 128} : act_params(a.mmA);
    const float a_scale = ap.scale;
    const int corr = PS ? 1 : 128 - ap.zp;
    for (int i = tid; i < S8_NF; i += S8_NT) {      // per-feature constants with the tensor-wide ones folded in: s = a_scale * w_scale[n], crz = (128 - a_zp) * (rowsum_w[n] - K z[n]) (PS: unfolded, the tile's own parameters join in the epilogue)
        c_ws[i] = a_scale * a.wscale[nbase + i]; c_rz[i] = corr * a.rsz[nbase + i]; c_b[i] = a.bias[nbase + i]; c_zw[i] = a.zw ? a.zw[nbase + i] : 0;
        if (EPI == SEPI_RESID_LN) { c_g[i] = a.gamma[i]; c_be[i] = a.beta[i]; }
    }
    float o_inv = 1.0f, o_zpf = 0.0f;
    if (EPI == SEPI_GELU_QUANT) { const ActQ op = act_params(a.mmO); o_inv = 1.0f / op.scale; o_zpf = (float)op.zp; }
    const bool has_zw = ZW;
    float xmax = -__builtin_inff(), xl = -__builtin_inff(), xr = __builtin_inff();      // SEPI_GELU_RANGE: largest pre-activation, nearest to gelu's argmin from the left / right

    uint32_t srcoff[NPC];
#pragma unroll
    for (int i = 0; i < NPC; ++i) {
        const int p = i * 256 + (tid & 255);
        const int row = p / 24, slot = p % 24;
        const int c = (slot & ~7) | ((slot & 7) ^ ((row >> 1) & 7));      // key (row >> 1) & 7: conflict-free for the lane groups of ds_read_b128 at a 384-byte pitch (see i8_stream_gelu_kernel)
        srcoff[i] = (uint32_t)(row * S8_PITCH + c * 16);
    }
    const unsigned char *xb = reinterpret_cast<const unsigned char *>(a.XQ);
    const uint32_t wave_lds = smem_lds + (uint32_t)(wave & 3) * 1024u;
    const int n_tiles = (M + S8_TR - 1) / S8_TR;         // the last tile may read up to 63 rows past M (the buffer is padded); they are not stored
    int t = worker;
#pragma unroll
    for (int b = 0; b < PF; ++b) {
        const int tt = t + b * n_workers < n_tiles ? t + b * n_workers : (t < n_tiles ? t : 0);
        const unsigned char *src = uniform_ptr(xb + (size_t)tt * S8_TILE);
        if (ZW && wave == 0) glds16(uniform_ptr(reinterpret_cast<const unsigned char *>(a.rsA + (size_t)tt * S8_TR)), (uint32_t)lane * 16u, smem_lds + (uint32_t)S8_RS + (uint32_t)b * 1024u);      // the tile's row sums travel with it (see i8_stream_gelu_kernel)
        if (wave < 4) {
#pragma unroll
            for (int i = 0; i < NPC; ++i) glds16(src, srcoff[i], wave_lds + b * S8_TILE + i * 4096);
        }
    }
    // resident weight fragments (A operand: row = feature l31 of the block, bytes 32 ks + 16 hi ..)
    i32x4q bw[KS];
    {
        const i32x4q *wp = reinterpret_cast<const i32x4q *>(a.Wp) + (size_t)nblk * KS * 64 + lane;
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) bw[ks] = wp[ks * 64];
    }
    int aoff[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) aoff[j] = l31 * S8_PITCH + (((2 * j + hi) ^ ((l31 >> 1) & 7)) << 4);

    uint32_t klo = 0xFFFFFFFFu, khi = 0u;
    float *red = reinterpret_cast<float *>(smem + S8_RED);
    unsigned char *scr = smem + S8_SCR + wave * 4096;

    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_waitcnt(0x0F70);   // vmcnt(0) for hipcc's own loads (weights): not re-waited inside the loop
    lds_barrier();
    const i32x16l zero16 = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    i32x16l acc0 = zero16, acc1 = zero16;
    uint32_t cur = 0;
    int it = 0, t_prev = 0;
    auto store_out_tile = [&](int tile, int buf) {        // SEPI_GELU_QUANT: rows of the finished output tile -> HBM, 16 B per lane, 384 B per token
        const unsigned char *ot = smem + S8_SCR + buf * S8_TILE;
#pragma unroll
        for (int pass = 0; pass < 2; ++pass) {
            const int p = pass * S8_NT + tid;
            const int row = p / 24, c = p % 24;
            const u32x4qq v4 = *reinterpret_cast<const u32x4qq *>(ot + row * S8_PITCH + (((c & ~7) | ((c & 7) ^ (row & 7))) << 4));
            const int m = tile * S8_TR + row;
            if (m < M) *reinterpret_cast<u32x4qq *>(a.out_q + (size_t)m * a.N + nbase + c * 16) = v4;
        }
    };
    for (; t < n_tiles; t += n_workers, ++it) {
        if (EPI == SEPI_GELU_QUANT) { if (it > 0) store_out_tile(t_prev, (it - 1) & 1); }
        float t_scale = 1.0f; int t_corr = 1, t_slot = 0;              // PS: this tile's DynamicQuantizeLinear parameters
        if (PS) { t_slot = ps_slot(t * S8_TR, a.mm_rows); const ActQ tp = act_params(a.mmA + 2 * t_slot); t_scale = tp.scale; t_corr = 128 - tp.zp; }
        const unsigned char *buf = smem + cur * S8_TILE;
        const uint32_t pfb = cur + PF >= S8_NBUF ? cur + PF - S8_NBUF : cur + PF;
        const int pt = t + PF * n_workers < n_tiles ? t + PF * n_workers : t;
        const unsigned char *psrc = uniform_ptr(xb + (size_t)pt * S8_TILE);
        const uint32_t pdst = (uint32_t)__builtin_amdgcn_readfirstlane((int)(wave_lds + pfb * S8_TILE));
        if (ZW && wave == 0) glds16(uniform_ptr(reinterpret_cast<const unsigned char *>(a.rsA + (size_t)pt * S8_TR)), (uint32_t)lane * 16u, (uint32_t)__builtin_amdgcn_readfirstlane((int)(smem_lds + (uint32_t)S8_RS + pfb * 1024u)));      // (older than this iteration's tile pieces: the counted wait below covers it)
        i32x4q ring[RING];
        auto rd = [&](int st) {
            const int rb = st / KS, ks = st % KS;
            ring[st % RING] = *reinterpret_cast<const i32x4q *>(buf + rb * 32 * S8_PITCH + aoff[ks & 3] + (ks >> 2) * 128);
        };
#pragma unroll
        for (int st = 0; st < D; ++st) rd(st);
#pragma unroll
        for (int st = 0; st < KS; ++st) {
            rd(st + D);
            acc0 = __builtin_amdgcn_mfma_i32_32x32x32_i8(bw[st], ring[st % RING], st == 0 ? zero16 : acc0, 0, 0, 0);
            if ((st & 3) == 2 && (st >> 2) < NPC) { if (wave < 4) glds16(psrc, srcoff[st >> 2], pdst + (st >> 2) * 4096); }
            __builtin_amdgcn_sched_barrier(0);
        }
#pragma unroll
        for (int st = KS; st < NS; ++st) {
            if (st + D < NS) rd(st + D);
            acc1 = __builtin_amdgcn_mfma_i32_32x32x32_i8(bw[st - KS], ring[st % RING], st == KS ? zero16 : acc1, 0, 0, 0);
            if ((st & 3) == 2 && (st >> 2) < NPC) { if (wave < 4) glds16(psrc, srcoff[st >> 2], pdst + (st >> 2) * 4096); }
            __builtin_amdgcn_sched_barrier(0);
        }
        // the next tile must have landed before the barrier: counted wait before this tile's own loads / stores are issued (see
        // gemm_k384_stream_kernel: everything younger than that tile's DMA is this tile's NPC pieces plus older, finished work)
        if (wave < 4) asm volatile("s_waitcnt vmcnt(%0)" ::"i"(NPC) : "memory");

        // ---- epilogue: lane = token (blk * 32 + l31 of the tile), registers = features 32 wave + f(r, hi); one 32-token block at a time
#pragma unroll
        for (int blk = 0; blk < 2; ++blk) {
            const i32x16l &acc = blk ? acc1 : acc0;
            const int m = t * S8_TR + blk * 32 + l31;
            const int mc = m < M ? m : M - 1;
            const bool valid = m < M;
            const int rsa = has_zw ? reinterpret_cast<const int32_t *>(smem + S8_RS)[cur * 256 + blk * 32 + l31] : 0;
            if (EPI == SEPI_GELU_RANGE) {
                if (valid) {
#pragma unroll
                    for (int g = 0; g < 4; ++g) {
                        const int nl = wave * 32 + 8 * g + 4 * hi;
                        const f32x4q ws = *reinterpret_cast<const f32x4q *>(c_ws + nl), b4 = *reinterpret_cast<const f32x4q *>(c_b + nl);
                        const i32x4q rz = *reinterpret_cast<const i32x4q *>(c_rz + nl), z4 = *reinterpret_cast<const i32x4q *>(c_zw + nl);
#pragma unroll
                        for (int e = 0; e < 4; ++e) {
                            const float x = (float)(acc[4 * g + e] + rz[e] - (ZW ? __mul24(z4[e], rsa) : 0)) * ws[e] + b4[e];
                            xmax = fmaxf(xmax, x);
                            xl = fmaxf(xl, x <= GELU_ARGMIN ? x : -__builtin_inff());
                            xr = fminf(xr, x >= GELU_ARGMIN ? x : __builtin_inff());
                        }
                    }
                }
            } else if (EPI == SEPI_GELU_QUANT) {
                unsigned char *ot = smem + S8_SCR + (it & 1) * S8_TILE + (blk * 32 + l31) * S8_PITCH;
                int ssum = 0;
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const int nl = wave * 32 + 8 * g + 4 * hi;
                    const f32x4q ws = *reinterpret_cast<const f32x4q *>(c_ws + nl), b4 = *reinterpret_cast<const f32x4q *>(c_b + nl);
                    const i32x4q rz = *reinterpret_cast<const i32x4q *>(c_rz + nl), z4 = *reinterpret_cast<const i32x4q *>(c_zw + nl);
                    uint32_t pk = 0;
#pragma unroll
                    for (int e2 = 0; e2 < 2; ++e2) {
                        f32x2q x2;
                        x2.x = (float)(acc[4 * g + 2 * e2] + rz[2 * e2] - (ZW ? __mul24(z4[2 * e2], rsa) : 0)) * ws[2 * e2] + b4[2 * e2];
                        x2.y = (float)(acc[4 * g + 2 * e2 + 1] + rz[2 * e2 + 1] - (ZW ? __mul24(z4[2 * e2 + 1], rsa) : 0)) * ws[2 * e2 + 1] + b4[2 * e2 + 1];
                        const f32x2q v2 = gelu_i8x2(x2);
                        // q = saturate(rint(v / scale) + zp) with v / scale taken as v * (1 / scale): the two differ in the last place at most, which
                        // moves a byte only when v / scale sits within 1e-7 of a rounding boundary -- the size of gelu_i8's own error
                        const float q0 = __builtin_rintf(v2.x * o_inv) + o_zpf, q1 = __builtin_rintf(v2.y * o_inv) + o_zpf;
                        pk = pack_u8(pk, q0, 2 * e2);
                        pk = pack_u8(pk, q1, 2 * e2 + 1);
                        if (ZW) ssum += (int)fminf(fmaxf(q0, 0.0f), 255.0f) + (int)fminf(fmaxf(q1, 0.0f), 255.0f) - 256;      // row sums of the stored bytes: only with weight zero points
                    }
                    pk ^= 0x80808080u;
                    const int c = nl >> 4;
                    *reinterpret_cast<uint32_t *>(ot + (((c & ~7) | ((c & 7) ^ (l31 & 7))) << 4) + (nl & 15)) = pk;      // (32 + l31) & 7 == l31 & 7
                }
                if (a.rs_out) {
                    ssum += __shfl_xor(ssum, 32);
                    if (hi == 0 && valid) atomicAdd(a.rs_out + m, ssum);
                }
            } else {
                // + residual, then LayerNorm over the 384 features of a token: 16 here, 16 in the other half-wave, the rest in the other 11 waves
                float v[16];
                float s = 0.0f;
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const int nl = wave * 32 + 8 * g + 4 * hi;
                    const f32x4q ws = *reinterpret_cast<const f32x4q *>(c_ws + nl), b4 = *reinterpret_cast<const f32x4q *>(c_b + nl);
                    const i32x4q rz = *reinterpret_cast<const i32x4q *>(c_rz + nl), z4 = *reinterpret_cast<const i32x4q *>(c_zw + nl);
                    const f32x4q r4 = *reinterpret_cast<const f32x4q *>(a.resid + (size_t)mc * S8_NF + nl);
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        if (PS) v[4 * g + e] = (float)(acc[4 * g + e] + __mul24(t_corr, rz[e]) - (ZW ? __mul24(z4[e], rsa) : 0)) * (t_scale * ws[e]) + b4[e] + r4[e];
                        else v[4 * g + e] = (float)(acc[4 * g + e] + rz[e] - (ZW ? __mul24(z4[e], rsa) : 0)) * ws[e] + b4[e] + r4[e];
                        s += v[4 * g + e];
                    }
                }
                float *rs_ = red + blk * 768, *rq_ = red + 384 + blk * 768;      // [12 waves][32 tokens] sums | squares of this block
                s += __shfl_xor(s, 32);
                if (hi == 0) rs_[wave * 32 + l31] = s;
                lds_barrier();
                float mean = 0.0f;
#pragma unroll
                for (int w = 0; w < 12; ++w) mean += rs_[w * 32 + l31];
                mean *= (1.0f / (float)S8_NF);
                float q = 0.0f;
#pragma unroll
                for (int r = 0; r < 16; ++r) { v[r] -= mean; q += v[r] * v[r]; }
                q += __shfl_xor(q, 32);
                if (hi == 0) rq_[wave * 32 + l31] = q;
                lds_barrier();
                float var = 0.0f;
#pragma unroll
                for (int w = 0; w < 12; ++w) var += rq_[w * 32 + l31];
                const float inv = 1.0f / sqrtf(var * (1.0f / (float)S8_NF) + a.eps);
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const int nl = wave * 32 + 8 * g + 4 * hi;
                    const f32x4q g4 = *reinterpret_cast<const f32x4q *>(c_g + nl), be4 = *reinterpret_cast<const f32x4q *>(c_be + nl);
                    f32x4q ov;
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        ov[e] = v[4 * g + e] * inv * g4[e] + be4[e];
                        if (valid) { const uint32_t kk = order_key(ov[e]); klo = min(klo, kk); khi = max(khi, kk); }
                    }
                    *reinterpret_cast<f32x4q *>(scr + l31 * 128 + (((2 * g + hi) ^ (l31 & 7)) << 4)) = ov;
                }
#pragma unroll
                for (int h = 0; h < 4; ++h) {
                    const int tl = h * 8 + (lane >> 3), ch = lane & 7;
                    const f32x4q v4 = *reinterpret_cast<const f32x4q *>(scr + tl * 128 + ((ch ^ (tl & 7)) << 4));
                    const int mt = t * S8_TR + blk * 32 + tl;
                    if (mt < M) *reinterpret_cast<f32x4q *>(a.out_f + (size_t)mt * S8_NF + wave * 32 + ch * 4) = v4;
                }
            }
        }
        if (PS && EPI == SEPI_RESID_LN && a.mm_out) {      // the tile's output range joins its sequence's
            for (int ofs = 32; ofs > 0; ofs >>= 1) { klo = min(klo, (uint32_t)__shfl_xor((int)klo, ofs)); khi = max(khi, (uint32_t)__shfl_xor((int)khi, ofs)); }
            if (lane == 0 && klo <= khi) {
                uint32_t *mo = a.mm_out + 2 * t_slot;
                if (klo < __atomic_load_n(mo, __ATOMIC_RELAXED)) atomicMin(mo, klo);
                if (khi > __atomic_load_n(mo + 1, __ATOMIC_RELAXED)) atomicMax(mo + 1, khi);
            }
            klo = 0xFFFFFFFFu; khi = 0u;
        }
        __builtin_amdgcn_s_waitcnt(0xC07F);   // lgkmcnt(0)
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
        cur = cur + 1 == S8_NBUF ? 0 : cur + 1;
        t_prev = t;
    }
    if (EPI == SEPI_GELU_QUANT) { if (it > 0) store_out_tile(t_prev, (it - 1) & 1); }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (EPI == SEPI_GELU_RANGE && a.mm_out) {       // mm_out = {key of max{x <= x*}, key of min{x >= x*}, key of max x}: gelu_range_finalize_kernel turns them into the range
        for (int ofs = 32; ofs > 0; ofs >>= 1) { xmax = fmaxf(xmax, __shfl_xor(xmax, ofs)); xl = fmaxf(xl, __shfl_xor(xl, ofs)); xr = fminf(xr, __shfl_xor(xr, ofs)); }
        if (lane == 0) {
            if (xl > -__builtin_inff()) { const uint32_t k = order_key(xl); if (k > __atomic_load_n(a.mm_out, __ATOMIC_RELAXED)) atomicMax(a.mm_out, k); }
            if (xr < __builtin_inff()) { const uint32_t k = order_key(xr); if (k < __atomic_load_n(a.mm_out + 1, __ATOMIC_RELAXED)) atomicMin(a.mm_out + 1, k); }
            if (xmax > -__builtin_inff()) { const uint32_t k = order_key(xmax); if (k > __atomic_load_n(a.mm_out + 2, __ATOMIC_RELAXED)) atomicMax(a.mm_out + 2, k); }
        }
    }
    if (!PS && EPI == SEPI_RESID_LN && a.mm_out) {
        for (int ofs = 32; ofs > 0; ofs >>= 1) { klo = min(klo, (uint32_t)__shfl_xor((int)klo, ofs)); khi = max(khi, (uint32_t)__shfl_xor((int)khi, ofs)); }
        if (lane == 0) {
            if (klo < __atomic_load_n(a.mm_out, __ATOMIC_RELAXED)) atomicMin(a.mm_out, klo);
            if (khi > __atomic_load_n(a.mm_out + 1, __ATOMIC_RELAXED)) atomicMax(a.mm_out + 1, khi);
        }
    }
}

// ---- the two FFN-up passes (range, then quantised bytes) with the epilogue of one token block running under the MFMAs of the next ---------------
// i8_stream_kernel<SEPI_GELU_*> computes a tile (24 dependent MFMAs per wave) and then post-processes it (GELU and quantisation: ~450 VALU
// instructions per wave and tile); its twelve waves meet at a barrier per tile, so all of them are in the matrix phase together and in the VALU phase
// together: measured 1.38 ms (bytes) / 0.84 ms (range) per layer where the matrix work is 0.28 ms and the VALU work 0.66 / 0.33 ms. Here the two
// accumulators alternate roles half a tile apart -- while block 0 of tile t accumulates, block 1 of tile t - 1 is post-processed from the other
// accumulator, then block 1 of tile t under the epilogue of its block 0 -- so every wave always has three MFMAs and a few dozen independent VALU
// instructions in the same scheduling region. Same tiles, same DMA ring, same barrier per tile; the output bytes of a tile are complete one
// iteration later, so the byte pass rotates three output tiles in LDS instead of two.
constexpr int S8G_LDS_RANGE = S8_RED, S8G_LDS_QUANT = S8_RED + 3 * S8_TILE;
// PS: one range per sequence (SHODH_QUANT_SCOPE_PER_TEXT; a.mm_rows = 128 or 256 rows per sequence, whole tiles). A worker takes whole SEQUENCES (its
// tiles one after the other, then the sequence n_workers further on), so that
//  * the DynamicQuantizeLinear parameters of its sequences -- input scale / correction, and for the byte pass the output scale / zero point -- are
//    computed once, at kernel start, into an LDS table (one entry per sequence of the worker; a global load of the range keys inside the pipelined loop
//    is waited for with a vmcnt that also covers the tile DMA in flight: the first form of this kernel, with such a load per tile and a load + atomics
//    per token block for the range, took 2.2 ms per layer against the batch scope's 0.65);
//  * the tile's parameters join in the epilogue (one more packed multiply per pair of values; the integer term as a fused multiply-add, still exact);
//  * the range pass keeps its three trackers in the wave across the tiles of a sequence and hands them over at the sequence's LAST token block: twelve
//    waves -> LDS -> wave 0 -> three unconditional atomics per sequence and feature group (stats [sequences][4]).
constexpr int S8G_PS_TAB = 128, S8G_PS_EXTRA = S8G_PS_TAB * 16 + 12 * 16;      // LDS behind the kernel's own: the table, then [12 waves][4] words of range hand-over
// ZW: 0 no weight zero points; 1 zero points, integer epilogue (an integer multiply + subtract + add per value before the conversion: the FFN-up passes are
// VALU-bound, 1.27 -> 1.90 and 0.75 -> 1.32 ms per layer in the per-text scope); 2 zero points whose terms provably stay below 2^24 for THIS tensor
// (QWeight::zw_bound, from the tensor's own bytes at load time): acc, corr * rowsum and zw * rowsum_a are then exactly representable floats and so is
// every partial sum, so float(acc) + corr * rsz - zw * rsa is the same number -- two packed fused multiply-adds per pair of values.
template <bool QUANT, int ZW, bool PS = false>
__global__ __launch_bounds__(768, 3) void i8_stream_gelu_kernel(const S8Args a) {
    constexpr int KS = S8_KS, NS = 2 * KS, D = (ZW == 2 ? 2 : 3), RING = 4, PF = S8_NBUF - 1, NPC = 2, OUT0 = S8_RED;      // (fragments three steps ahead: with the epilogue between the MFMAs a step is > 100 cycles; two in the float zero-point form, whose epilogue holds more registers: 9 / 14 -> 4 spilled in the quantising pass, 40.8 -> 40.3 ms per forward)
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const uint32_t smem_lds = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) unsigned char *)smem;
    const int tid = threadIdx.x, lane = tid & 63, hi = lane >> 5, l31 = lane & 31;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int M = a.M;
    const int n_workers = gridDim.x / a.n_groups;
    const int xcd = blockIdx.x & 7, rr = blockIdx.x >> 3;
    const int grp = rr % a.n_groups, worker = (rr / a.n_groups) * 8 + xcd;
    const int nbase = grp * S8_NF;
    const int nblk = grp * 12 + wave;

    float *c_ws = reinterpret_cast<float *>(smem + S8_CONST);
    int32_t *c_rz = reinterpret_cast<int32_t *>(c_ws + S8_NF);
    float *c_b = reinterpret_cast<float *>(c_rz + S8_NF);
    int32_t *c_zw = reinterpret_cast<int32_t *>(c_b + S8_NF);
    // With weight zero points every token needs the row sum of its quantised bytes. A global load for it inside the pipelined regions would be waited
    // for with the compiler's own vmcnt, which knows nothing of the tile DMA issued just before it and so waits for THAT (an HBM round trip per token
    // block: the zero-point variants ran 1.61 / 0.84 ms against 1.17 / 0.66). The 64 sums of a tile travel with the tile instead: one more DMA
    // piece from wave 0 (a whole 1-KiB piece, of which the first 256 bytes matter: no divergent branch around the asm) into a slot behind the
    // constants (the LayerNorm gains' place in i8_stream_kernel), read back with ds_read_b32.
    int32_t *c_rs = c_zw + S8_NF;                                    // [S8_NBUF][256], 64 used
    const uint32_t rs_lds = smem_lds + (uint32_t)S8_CONST + 4u * S8_NF * 4u;
    const ActQ ap = PS ? ActQ{1.0f, 127} : act_params(a.mmA);       // (PS: scale 1 and correction factor 1 = the constants stay unfolded)
    // PS: sequence-major tile order and the parameter table
    const int tps_log2 = PS ? (a.mm_rows == 256 ? 2 : 1) : 0;       // tiles per sequence: 4 or 2
    const int nseq_all = PS ? M / a.mm_rows : 0;
    f32x4q *ps_tab = reinterpret_cast<f32x4q *>(smem + (QUANT ? S8G_LDS_QUANT : S8G_LDS_RANGE));
    uint32_t *ps_red = reinterpret_cast<uint32_t *>(ps_tab + S8G_PS_TAB);
    if (PS) {
        for (int q = tid; q < S8G_PS_TAB; q += S8_NT) {
            const int sq = worker + q * n_workers;
            if (sq < nseq_all) {
                const ActQ tp = act_params(a.mmA + 2 * sq);
                f32x4q ent; ent[0] = tp.scale; ent[1] = (float)(128 - tp.zp); ent[2] = 1.0f; ent[3] = 0.0f;
                if (QUANT) { const ActQ op = act_params(a.mmO + 2 * sq); ent[2] = 1.0f / op.scale; ent[3] = (float)op.zp; }
                ps_tab[q] = ent;
            }
        }
    }
    const float a_scale = ap.scale;
    const int corr = 128 - ap.zp;
    // Without weight zero points the integer term (128 - a_zp) * rowsum_w[n] is added as a FLOAT after the conversion: |acc| <= 384 * 128 * 127 and
    // |term| <= 128 * 384 * 128 are integers below 2^23, so is their sum -- the float addition is exact and equals float(acc + term) (one conversion and
    // half a packed add per value instead of an integer add and a conversion). With zero points the terms outgrow that and stay integer.
    for (int i = tid; i < S8_NF; i += S8_NT) {
        c_ws[i] = a_scale * a.wscale[nbase + i]; c_b[i] = a.bias[nbase + i];
        if (ZW == 2) reinterpret_cast<float *>(c_zw)[i] = (float)a.zw[nbase + i]; else c_zw[i] = a.zw ? a.zw[nbase + i] : 0;
        if (ZW == 1) c_rz[i] = corr * a.rsz[nbase + i]; else reinterpret_cast<float *>(c_rz)[i] = (float)(corr * a.rsz[nbase + i]);
    }
    float o_inv = 1.0f, o_zpf = 0.0f;
    if (QUANT && !PS) { const ActQ op = act_params(a.mmO); o_inv = 1.0f / op.scale; o_zpf = (float)op.zp; }
    // range pass: the wave's running {distance below x* of the nearest value left of it, distance above of the nearest NEGATIVE value right of it} as
    // differences of IEEE bit patterns (see epi_chunk), and the largest value
    constexpr uint32_t XSB = 0xBF40756Au;                            // bits of GELU_ARGMIN = -0.7517915964f
    static_assert(__builtin_bit_cast(uint32_t, GELU_ARGMIN) == XSB, "XSB holds the bit pattern of GELU_ARGMIN");
    uint32_t w_dl = 0xFFFFFFFFu, w_dr = 0xFFFFFFFFu;
    float xmax = -__builtin_inff();

    // the tile DMA is shared by all twelve waves (two 1-KiB pieces each; i8_stream_kernel lets waves 0-3 issue six each behind a branch, which would
    // cut the scheduling regions below in two)
    uint32_t srcoff[NPC];
#pragma unroll
    for (int i = 0; i < NPC; ++i) {
        const int p = i * S8_NT + tid;
        const int row = p / 24, slot = p % 24;
        const int c = (slot & ~7) | ((slot & 7) ^ ((row >> 1) & 7));      // key (row >> 1) & 7: see aoff
        srcoff[i] = (uint32_t)(row * S8_PITCH + c * 16);
    }
    const unsigned char *xb = reinterpret_cast<const unsigned char *>(a.XQ);
    const uint32_t wave_lds = smem_lds + (uint32_t)wave * 1024u;
    const int n_tiles = (M + S8_TR - 1) / S8_TR;
    // the tile of iteration i of this worker (n_tiles and beyond: none). Batch scope: tiles worker, worker + n_workers, ...; PS: the tiles of sequence
    // worker, then of sequence worker + n_workers, ...
    auto tile_at = [&](int i) -> int {
        if (!PS) return worker + i * n_workers;
        const int sq = worker + (i >> tps_log2) * n_workers;
        return sq < nseq_all ? (sq << tps_log2) + (i & ((1 << tps_log2) - 1)) : n_tiles;
    };
    int t = tile_at(0);
#pragma unroll
    for (int b = 0; b < PF; ++b) {
        const int tt = tile_at(b) < n_tiles ? tile_at(b) : (t < n_tiles ? t : 0);
        const unsigned char *src = uniform_ptr(xb + (size_t)tt * S8_TILE);
#pragma unroll
        for (int i = 0; i < NPC; ++i) glds16(src, srcoff[i], wave_lds + b * S8_TILE + i * 12288);
        if (ZW && wave == 0) glds16(uniform_ptr(reinterpret_cast<const unsigned char *>(a.rsA + (size_t)tt * S8_TR)), (uint32_t)lane * 16u, rs_lds + (uint32_t)b * 1024u);
    }
    i32x4q bw[KS];
    {
        const i32x4q *wp = reinterpret_cast<const i32x4q *>(a.Wp) + (size_t)nblk * KS * 64 + lane;
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) bw[ks] = wp[ks * 64];
    }
    // A-fragment reads: ds_read_b128 is served in four groups of sixteen lanes -- {0-3, 12-15, 20-27}, {4-11, 16-19, 28-31} and the same + 32 -- over
    // 64 banks (256 B). Rows are 384 B apart, so a row's 128-byte segment starts at bank 0 or 32 by the row's parity, and within each group the eight
    // even and the eight odd rows must take eight different 16-byte slots: the XOR key is (row >> 1) & 7. (With row & 7, i8_stream_kernel's key, rows
    // 12 and 20 -- same group, same parity, same key -- collide: SQ_LDS_BANK_CONFLICT was a third of SQ_LDS_IDX_ACTIVE.)
    int aoff[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) aoff[j] = l31 * S8_PITCH + (((2 * j + hi) ^ ((l31 >> 1) & 7)) << 4);

    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_waitcnt(0x0F70);
    lds_barrier();
    const i32x16l zero16 = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    i32x16l acc0 = zero16, acc1 = zero16;
    uint32_t cur = 0;
    int it = 0, t_p1 = 0, t_p2 = 0;                                 // the tiles of the previous two iterations
    int rsa_b1 = 0;
    auto store_out_tile = [&](int tile, int buf) {
        const unsigned char *ot = smem + OUT0 + buf * S8_TILE;
#pragma unroll
        for (int pass = 0; pass < 2; ++pass) {
            const int p = pass * S8_NT + tid;
            const int row = p / 24, c = p % 24;
            const u32x4qq v4 = *reinterpret_cast<const u32x4qq *>(ot + row * S8_PITCH + (((c & ~7) | ((c & 7) ^ (row & 7))) << 4));
            const int m = tile * S8_TR + row;
            if (m < M) *reinterpret_cast<u32x4qq *>(a.out_q + (size_t)m * a.N + nbase + c * 16) = v4;
        }
    };
    // four values (features 32 wave + 8 g + 4 hi ..) of token (tile, blk, l31) from accumulator registers 4 g .. 4 g + 3
    struct BlockAcc { uint32_t dl, dr; float mx; int ssum; };       // per token block: range trackers / row sum of the bytes
    struct TileQ { float as; int corr; float corr_f; float o_inv, o_zpf; int slot; };      // PS: the DynamicQuantizeLinear parameters of a tile's sequence (input: as / corr, output of the byte pass: o_inv / o_zpf)
    auto tile_params = [&](int i) {                                   // of iteration i's tile
        TileQ q = {1.0f, 1, 1.0f, o_inv, o_zpf, 0};
        if (PS) {
            const int ql = (i >> tps_log2) & (S8G_PS_TAB - 1);
            const f32x4q ent = ps_tab[ql];
            q.slot = worker + (i >> tps_log2) * n_workers;
            // (wave-uniform values: through readfirstlane they live in scalar registers -- two tiles' parameters in vector registers made the zero-point variants spill)
            auto uni = [](float v) { return __uint_as_float((uint32_t)__builtin_amdgcn_readfirstlane((int)__float_as_uint(v))); };
            q.as = uni(ent[0]); q.corr_f = uni(ent[1]); q.corr = (int)q.corr_f; q.o_inv = uni(ent[2]); q.o_zpf = uni(ent[3]);
        }
        return q;
    };
    auto epi_chunk = [&](const i32x16l &acc, int g, int rsa, unsigned char *orow, BlockAcc &ba, const TileQ &tq) {
        const int nl = wave * 32 + 8 * g + 4 * hi;
        const f32x4q ws = *reinterpret_cast<const f32x4q *>(c_ws + nl), b4 = *reinterpret_cast<const f32x4q *>(c_b + nl);
        f32x2q x2[2];
        if (ZW == 2) {
            const f32x2q nrsa = (f32x2q)(-(float)rsa);
#pragma unroll
            for (int e2 = 0; e2 < 2; ++e2) {
                const f32x2q r = *reinterpret_cast<const f32x2q *>(reinterpret_cast<const float *>(c_rz) + nl + 2 * e2), z = *reinterpret_cast<const f32x2q *>(reinterpret_cast<const float *>(c_zw) + nl + 2 * e2);
                f32x2q f, w, b;
#pragma unroll
                for (int e = 0; e < 2; ++e) { f[e] = (float)acc[4 * g + 2 * e2 + e]; w[e] = ws[2 * e2 + e]; b[e] = b4[2 * e2 + e]; }
                f32x2q t = PS ? __builtin_elementwise_fma((f32x2q)tq.corr_f, r, f) : f + r;      // integers below 2^24 at every step (see the template comment): exact
                t = __builtin_elementwise_fma(nrsa, z, t);
                x2[e2] = t * (PS ? w * tq.as : w) + b;
            }
        } else if (ZW) {
            const i32x4q rz = *reinterpret_cast<const i32x4q *>(c_rz + nl), z4 = *reinterpret_cast<const i32x4q *>(c_zw + nl);
#pragma unroll
            for (int e2 = 0; e2 < 2; ++e2) {
                f32x2q f, w, b;
#pragma unroll
                for (int e = 0; e < 2; ++e) { f[e] = (float)(acc[4 * g + 2 * e2 + e] + (PS ? __mul24(tq.corr, rz[2 * e2 + e]) : rz[2 * e2 + e]) - __mul24(z4[2 * e2 + e], rsa)); w[e] = ws[2 * e2 + e]; b[e] = b4[2 * e2 + e]; }
                if (PS) w = w * tq.as;
                x2[e2] = f * w + b;
            }
        } else {
            const f32x4q rzf = *reinterpret_cast<const f32x4q *>(reinterpret_cast<const float *>(c_rz) + nl);
#pragma unroll
            for (int e2 = 0; e2 < 2; ++e2) {
                f32x2q f, r, w, b;
#pragma unroll
                for (int e = 0; e < 2; ++e) { f[e] = (float)acc[4 * g + 2 * e2 + e]; r[e] = rzf[2 * e2 + e]; w[e] = ws[2 * e2 + e]; b[e] = b4[2 * e2 + e]; }
                if (PS) x2[e2] = __builtin_elementwise_fma((f32x2q)tq.corr_f, r, f) * (w * tq.as) + b;      // corr * rowsum and its sum with acc are integers below 2^24: the fused form is exact, like the folded one
                else x2[e2] = (f + r) * w + b;
            }
        }
        if (!QUANT) {
            // nearest to x* = GELU_ARGMIN (negative) from the left: x <= x* <=> bits(x) >= bits(x*) as unsigned (negative floats grow in magnitude with
            // their bits, everything non-negative lies below 0x80000000), and the nearest has the smallest bits: min of bits(x) - bits(x*), which wraps
            // to something huge for every other x. From the right only NEGATIVE values matter (gelu >= 0 from 0 on, and the range contains 0 anyway):
            // min of bits(x*) - bits(x); a non-negative x gives at least bits(x*) - 0x7F800000, more than any x in [x*, -0]. Two subtractions and
            // 1.5 three-operand min / max per value, exact -- the values come back as x = bits(x*) +- difference.
#pragma unroll
            for (int e2 = 0; e2 < 2; ++e2) {
                const uint32_t u0 = __float_as_uint(x2[e2][0]), u1 = __float_as_uint(x2[e2][1]);
                ba.dl = min(min(ba.dl, u0 - XSB), u1 - XSB);
                ba.dr = min(min(ba.dr, XSB - u0), XSB - u1);
                ba.mx = fmaxf(fmaxf(ba.mx, x2[e2][0]), x2[e2][1]);
            }
        } else {
            uint32_t pk = 0;
#pragma unroll
            for (int e2 = 0; e2 < 2; ++e2) {
                const f32x2q v2 = gelu_i8x2(x2[e2]);
                // q = saturate(round_half_even(v / scale + zp)): v / scale is taken as v * (1 / scale) and the zero point joins in the same fused
                // multiply-add; v_cvt_pk_u8_f32 rounds half to even and saturates. Against rint(v / scale) + zp this moves a byte only when
                // v / scale sits within ~1e-5 of a rounding boundary -- the size of gelu_i8's own error (1e-7 |x| / scale) -- and saves the
                // separate multiply, rint and add (five of the eight slots of a pair).
                const f32x2q t2 = __builtin_elementwise_fma(v2, (f32x2q)tq.o_inv, (f32x2q)tq.o_zpf);
                pk = pack_u8(pk, t2[0], 2 * e2);
                pk = pack_u8(pk, t2[1], 2 * e2 + 1);
            }
            pk ^= 0x80808080u;
            if (ZW) ba.ssum = __builtin_amdgcn_sdot4((int)pk, 0x01010101, ba.ssum, false);      // row sums of the stored bytes: only with weight zero points
            const int c = nl >> 4;
            *reinterpret_cast<uint32_t *>(orow + (((c & ~7) | ((c & 7) ^ (l31 & 7))) << 4) + (nl & 15)) = pk;
        }
    };
    auto commit_range = [&](uint32_t *st, uint32_t dl, uint32_t dr, float mx) {     // {nearest left of x*, nearest right of it, largest} -> the three order keys of `st` (see gelu_range_finalize_kernel)
        for (int ofs = 32; ofs > 0; ofs >>= 1) {
            mx = fmaxf(mx, __shfl_xor(mx, ofs));
            dl = min(dl, (uint32_t)__shfl_xor((int)dl, ofs)); dr = min(dr, (uint32_t)__shfl_xor((int)dr, ofs));
        }
        if (lane == 0) {
            if (dl <= 0xFF800000u - XSB) { const uint32_t k = order_key(__uint_as_float(XSB + dl)); if (k > __atomic_load_n(st, __ATOMIC_RELAXED)) atomicMax(st, k); }            // a finite x (or -inf) at or left of x*
            if (dr <= XSB - 0x80000000u) { const uint32_t k = order_key(__uint_as_float(XSB - dr)); if (k < __atomic_load_n(st + 1, __ATOMIC_RELAXED)) atomicMin(st + 1, k); }      // a negative x in [x*, -0]
            if (mx > -__builtin_inff()) { const uint32_t k = order_key(mx); if (k > __atomic_load_n(st + 2, __ATOMIC_RELAXED)) atomicMax(st + 2, k); }
        }
    };
    int ps_pending = -1;                                              // PS range pass: the sequence whose trackers lie in ps_red, to be committed by wave 0 behind the next barrier
    auto finish_block = [&](int m, bool valid, const BlockAcc &ba, const TileQ &tq, bool seq_end) {  // a finished token block: its trackers join the wave's (padding rows of the last tile do not), its row sums go out
        if (!QUANT) {
            w_dl = min(w_dl, valid ? ba.dl : 0xFFFFFFFFu);
            w_dr = min(w_dr, valid ? ba.dr : 0xFFFFFFFFu);
            xmax = fmaxf(xmax, valid ? ba.mx : -__builtin_inff());
            if (PS && seq_end) {                                      // (uniform) the sequence's last block: the wave's trackers -> LDS, and start afresh
                for (int ofs = 32; ofs > 0; ofs >>= 1) {
                    xmax = fmaxf(xmax, __shfl_xor(xmax, ofs));
                    w_dl = min(w_dl, (uint32_t)__shfl_xor((int)w_dl, ofs)); w_dr = min(w_dr, (uint32_t)__shfl_xor((int)w_dr, ofs));
                }
                if (lane == 0) { ps_red[wave * 4] = w_dl; ps_red[wave * 4 + 1] = w_dr; ps_red[wave * 4 + 2] = __float_as_uint(xmax); }
                w_dl = 0xFFFFFFFFu; w_dr = 0xFFFFFFFFu; xmax = -__builtin_inff();
                ps_pending = tq.slot;
            }
        } else if (ZW && a.rs_out) {
            int ssum = ba.ssum;
            ssum += __shfl_xor(ssum, 32);
            if (hi == 0 && valid) atomicAdd(a.rs_out + m, ssum);
        }
    };
    auto ps_commit = [&]() {                                          // behind a barrier: wave 0 folds the twelve waves' trackers of sequence ps_pending and sends them off
        if (PS && !QUANT && ps_pending >= 0) {
            if (wave == 0 && a.mm_out) {
                uint32_t dl = lane < 12 ? ps_red[lane * 4] : 0xFFFFFFFFu, dr = lane < 12 ? ps_red[lane * 4 + 1] : 0xFFFFFFFFu;
                float mx = lane < 12 ? __uint_as_float(ps_red[lane * 4 + 2]) : -__builtin_inff();
                for (int ofs = 8; ofs > 0; ofs >>= 1) {
                    mx = fmaxf(mx, __shfl_xor(mx, ofs));
                    dl = min(dl, (uint32_t)__shfl_xor((int)dl, ofs)); dr = min(dr, (uint32_t)__shfl_xor((int)dr, ofs));
                }
                uint32_t *st = a.mm_out + 4 * ps_pending;
                if (lane == 0) {
                    if (dl <= 0xFF800000u - XSB) atomicMax(st, order_key(__uint_as_float(XSB + dl)));
                    if (dr <= XSB - 0x80000000u) atomicMin(st + 1, order_key(__uint_as_float(XSB - dr)));
                    if (mx > -__builtin_inff()) atomicMax(st + 2, order_key(mx));
                }
            }
            ps_pending = -1;
        }
    };
    const BlockAcc ba_init = {0xFFFFFFFFu, 0xFFFFFFFFu, -__builtin_inff(), 0};
    TileQ tq_cur = tile_params(0), tq_p1 = tq_cur;                    // (not PS: the kernel-wide constants)
    const int tps_mask = (1 << tps_log2) - 1;
    for (; t < n_tiles; t = tile_at(++it)) {
        if (QUANT) { if (it > 1) store_out_tile(t_p2, (it - 2) % 3); }
        if (PS) { tq_p1 = tq_cur; tq_cur = tile_params(it); }
        const unsigned char *buf = smem + cur * S8_TILE;
        const uint32_t pfb = cur + PF >= S8_NBUF ? cur + PF - S8_NBUF : cur + PF;
        const int pt = tile_at(it + PF) < n_tiles ? tile_at(it + PF) : t;
        const unsigned char *psrc = uniform_ptr(xb + (size_t)pt * S8_TILE);
        const uint32_t pdst = (uint32_t)__builtin_amdgcn_readfirstlane((int)(wave_lds + pfb * S8_TILE));
        const int rsa_p1 = rsa_b1;                                  // row sums of the previous tile's block 1 (its slot is the one being refilled now: taken while it was this tile)
        rsa_b1 = ZW ? c_rs[cur * 256 + 32 + l31] : 0;
        if (ZW && wave == 0) glds16(uniform_ptr(reinterpret_cast<const unsigned char *>(a.rsA + (size_t)pt * S8_TR)), (uint32_t)lane * 16u, (uint32_t)__builtin_amdgcn_readfirstlane((int)(rs_lds + pfb * 1024u)));
        i32x4q ring[RING];
        auto rd = [&](int st) {
            const int rb = st / KS, ks = st % KS;
            ring[st % RING] = *reinterpret_cast<const i32x4q *>(buf + rb * 32 * S8_PITCH + aoff[ks & 3] + (ks >> 2) * 128);
        };
        auto mfma_step = [&](int st) {
            if (st + D < NS) rd(st + D);
            if (st < KS) acc0 = __builtin_amdgcn_mfma_i32_32x32x32_i8(bw[st], ring[st % RING], st == 0 ? zero16 : acc0, 0, 0, 0);
            else acc1 = __builtin_amdgcn_mfma_i32_32x32x32_i8(bw[st - KS], ring[st % RING], st == KS ? zero16 : acc1, 0, 0, 0);
            if (st == 2) glds16(psrc, srcoff[0], pdst);
            if (st == KS + 2) glds16(psrc, srcoff[1], pdst + 12288);
        };
#pragma unroll
        for (int st = 0; st < D; ++st) rd(st);
        // ---- phase 1: block 0 of this tile accumulates; block 1 of the previous tile is post-processed from acc1
        if (it == 0) {
#pragma unroll
            for (int st = 0; st < KS; ++st) { mfma_step(st); __builtin_amdgcn_sched_barrier(0); }
        } else {
            const int m = t_p1 * S8_TR + 32 + l31;
            const bool valid = m < M;
            const int rsa = rsa_p1;
            unsigned char *orow = smem + OUT0 + ((it - 1) % 3) * S8_TILE + (32 + l31) * S8_PITCH;
            BlockAcc ba = ba_init;
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                mfma_step(3 * g); mfma_step(3 * g + 1); mfma_step(3 * g + 2);
                epi_chunk(acc1, g, rsa, orow, ba, tq_p1);
#pragma unroll
                for (int j = 0; j < 3; ++j) { __builtin_amdgcn_sched_group_barrier(0x008, 1, 0); __builtin_amdgcn_sched_group_barrier(0x002, QUANT ? 20 : 10, 0); }
                __builtin_amdgcn_sched_barrier(0);
            }
            finish_block(m, valid, ba, tq_p1, ((it - 1) & tps_mask) == tps_mask);      // (block 1 of the sequence's last tile ends the sequence)
        }
        // ---- phase 2: block 1 accumulates; block 0 of this tile is post-processed from acc0
        {
            const int m = t * S8_TR + l31;
            const bool valid = m < M;
            const int rsa = ZW ? c_rs[cur * 256 + l31] : 0;
            unsigned char *orow = smem + OUT0 + (it % 3) * S8_TILE + l31 * S8_PITCH;
            BlockAcc ba = ba_init;
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                mfma_step(KS + 3 * g); mfma_step(KS + 3 * g + 1); mfma_step(KS + 3 * g + 2);
                epi_chunk(acc0, g, rsa, orow, ba, tq_cur);
#pragma unroll
                for (int j = 0; j < 3; ++j) { __builtin_amdgcn_sched_group_barrier(0x008, 1, 0); __builtin_amdgcn_sched_group_barrier(0x002, QUANT ? 20 : 10, 0); }
                __builtin_amdgcn_sched_barrier(0);
            }
            finish_block(m, valid, ba, tq_cur, false);
        }
        // the next tile must have landed before the barrier (counted wait: everything younger than that tile's DMA is this tile's NPC pieces)
        asm volatile("s_waitcnt vmcnt(%0)" ::"i"(NPC) : "memory");      // (wave 0's row-sum piece of this iteration was requested BEFORE the two tile pieces: it is older than the NPC allowed to stay out and simply lands an iteration early; a wave-dependent count behind a branch here made the register allocator spill the weight fragments)
        __builtin_amdgcn_s_waitcnt(0xC07F);   // lgkmcnt(0)
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
        ps_commit();
        cur = cur + 1 == S8_NBUF ? 0 : cur + 1;
        t_p2 = t_p1; t_p1 = t;
    }
    // ---- drain: the last tile's block 1, then the output tiles still in LDS
    if (it > 0) {
        if (QUANT) { if (it > 1) store_out_tile(t_p2, (it - 2) % 3); }
        const int m = t_p1 * S8_TR + 32 + l31;
        const bool valid = m < M;
        const int rsa = rsa_b1;
        unsigned char *orow = smem + OUT0 + ((it - 1) % 3) * S8_TILE + (32 + l31) * S8_PITCH;
        BlockAcc ba = ba_init;
#pragma unroll
        for (int g = 0; g < 4; ++g) epi_chunk(acc1, g, rsa, orow, ba, tq_cur);      // (the last tile this worker ran)
        finish_block(m, valid, ba, tq_cur, true);
        if (PS && !QUANT) {
            __builtin_amdgcn_s_waitcnt(0xC07F);
            __builtin_amdgcn_s_barrier();
            ps_commit();
        }
        if (QUANT) {
            __builtin_amdgcn_s_waitcnt(0xC07F);
            __builtin_amdgcn_s_barrier();
            store_out_tile(t_p1, (it - 1) % 3);
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (!QUANT && !PS && a.mm_out) commit_range(a.mm_out, w_dl, w_dr, xmax);      // mm_out = {key of max{x <= x*}, key of min{x >= x*}, key of max x}: gelu_range_finalize_kernel turns them into the range
}

// ---- tiled int8 GEMM over all 384 output features for K = 1536 (FFN down) + residual + LayerNorm ---------------------------------
// 128 tokens x 384 features per workgroup, 8 waves = 2 token halves x 4 feature quarters (2 x 3 MFMA blocks each), K tiles of 128 bytes,
// register-staged double-buffered LDS (rows of 128 B, 16-B chunks XOR-swizzled by (row >> 1) & 7). The weights are re-read per workgroup from
// L2 (576 KiB); the kernel is bound by its three HBM streams (quantised activations, residual, output: 4.6 KB per token).
constexpr int KT_TM = 128, KT_NF = 384, KT_KB = 128, KT_STAGE = (KT_TM + KT_NF) * KT_KB;      // 64 KiB per stage
constexpr int KT_CONST = 2 * KT_STAGE, KT_LDS = KT_CONST + 6 * KT_NF * 4;
// ZWK: the weight has zero points; the row sums of the activation bytes that term needs are formed here from the fragments the wave reads anyway
// (v_dot4 with ones: 32 per K tile and wave next to 24 MFMAs) -- the producer does not have to emit them (it did, with 48 atomic adds per token).
template <bool ZWK>
__global__ __launch_bounds__(512, 2) void i8_ktile_ln_kernel(const int8_t *__restrict__ A /* [M][K] signed storage */, const int32_t *__restrict__ rsA /* unused */, const uint32_t *__restrict__ mmA,
                                                             const int8_t *__restrict__ W /* [384][K] row-major */, const float *__restrict__ wscale, const int32_t *__restrict__ rsz,
                                                             const int32_t *__restrict__ zw, const float *__restrict__ bias, const float *resid, const float *__restrict__ gamma,
                                                             const float *__restrict__ beta, float eps, float *out_f /* may alias resid */, uint32_t *__restrict__ mm_out, int M, int K,
                                                             int mm_rows /* 0: one range per tensor; else rows per sequence (a multiple of 128): mmA / mm_out are [sequences][2] */) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int tid = threadIdx.x, lane = tid & 63, hi = lane >> 5, l31 = lane & 31;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = wave >> 2, wc = wave & 3;
    const int m0 = blockIdx.x * KT_TM;
#ifdef SHODH_KT_ABL    // diagnostic builds only (-DSHODH_KT_ABL=mask): 1 no W loads, 2 no A loads, 4 no residual loads, 8 no stores, 16 no MFMAs, 32 no LDS staging (results invalid)
    constexpr int abl = SHODH_KT_ABL;
#else
    constexpr int abl = 0;
#endif
    if (mm_rows) { const int slot = m0 / mm_rows; mmA += 2 * slot; if (mm_out) mm_out += 2 * slot; }
    float *c_ws = reinterpret_cast<float *>(smem + KT_CONST);
    int32_t *c_rz = reinterpret_cast<int32_t *>(c_ws + KT_NF);
    float *c_b = reinterpret_cast<float *>(c_rz + KT_NF);
    int32_t *c_zw = reinterpret_cast<int32_t *>(c_b + KT_NF);
    float *c_g = reinterpret_cast<float *>(c_zw + KT_NF), *c_be = c_g + KT_NF;
    for (int i = tid; i < KT_NF; i += 512) { c_ws[i] = wscale[i]; c_rz[i] = rsz[i]; c_b[i] = bias[i]; c_zw[i] = zw ? zw[i] : 0; c_g[i] = gamma[i]; c_be[i] = beta[i]; }
    const ActQ ap = act_params(mmA);
    const float a_scale = ap.scale;
    const int corr = 128 - ap.zp;
    // staging: A tile 128 rows x 8 chunks = 1024 chunks (2 per thread), W tile 384 x 8 = 3072 (6 per thread)
    u32x4qq pa[2], pw[6];
    const int8_t *ga[2], *gw[6];
    int la[2], lw[6];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int sidx = i * 512 + tid, row = sidx >> 3, c = sidx & 7;
        int ra = m0 + row; if (ra >= M) ra = M - 1;
        ga[i] = A + (size_t)ra * K + c * 16;
        la[i] = row * KT_KB + ((c ^ ((row >> 1) & 7)) << 4);
    }
#pragma unroll
    for (int i = 0; i < 6; ++i) {
        const int sidx = i * 512 + tid, row = sidx >> 3, c = sidx & 7;
        gw[i] = W + (size_t)row * K + c * 16;
        lw[i] = KT_TM * KT_KB + row * KT_KB + ((c ^ ((row >> 1) & 7)) << 4);
    }
    auto gload = [&](int kt) {
#pragma unroll
        for (int i = 0; i < 2; ++i) if (!(abl & 2) || kt == 0) pa[i] = *reinterpret_cast<const u32x4qq *>(ga[i] + kt * KT_KB);
#pragma unroll
        for (int i = 0; i < 6; ++i) if (!(abl & 1) || kt == 0) pw[i] = *reinterpret_cast<const u32x4qq *>(gw[i] + kt * KT_KB);
    };
    auto stage = [&](int buf) {
        unsigned char *base = smem + buf * KT_STAGE;
#pragma unroll
        for (int i = 0; i < 2; ++i) *reinterpret_cast<u32x4qq *>(base + la[i]) = pa[i];
#pragma unroll
        for (int i = 0; i < 6; ++i) *reinterpret_cast<u32x4qq *>(base + lw[i]) = pw[i];
    };
    i32x16l acc[3][2];      // [feature block j][token block i]
#pragma unroll
    for (int j = 0; j < 3; ++j)
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[j][i][r] = 0;
    int fo_a[2], fo_w[3];
    int rs_acc[2] = {0, 0};                    // ZWK: this lane's half of its tokens' row sums
    // key (row >> 1) & 7: rows of 128 B start at bank 0 or 32 by their parity, and the sixteen lanes of a ds_read_b128 group ({0-3, 12-15, 20-27}, ...) hold
    // eight even and eight odd rows whose (row >> 1) & 7 are all different (with row & 7, rows 12 and 20 collided: a quarter of the LDS cycles were conflicts)
    const int sw = (l31 >> 1) & 7;   // every fragment row of this lane has the same key (block bases are multiples of 32)
#pragma unroll
    for (int i = 0; i < 2; ++i) fo_a[i] = (wr * 64 + i * 32 + l31) * KT_KB;
#pragma unroll
    for (int j = 0; j < 3; ++j) fo_w[j] = KT_TM * KT_KB + (wc * 96 + j * 32 + l31) * KT_KB;
    const int nkt = K / KT_KB;
    gload(0);
    stage(0);
    __syncthreads();
    for (int kt = 0; kt < nkt; ++kt) {
        const int cur = kt & 1;
        if (kt + 1 < nkt) gload(kt + 1);
        const unsigned char *tb = smem + cur * KT_STAGE;
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            i32x4q fa[2], fw[3];
            const int co = ((ks * 2 + hi) ^ sw) << 4;
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                fa[i] = *reinterpret_cast<const i32x4q *>(tb + fo_a[i] + co);
                if (ZWK) {
#pragma unroll
                    for (int c = 0; c < 4; ++c) rs_acc[i] = __builtin_amdgcn_sdot4(fa[i][c], 0x01010101, rs_acc[i], false);
                }
            }
#pragma unroll
            for (int j = 0; j < 3; ++j) fw[j] = *reinterpret_cast<const i32x4q *>(tb + fo_w[j] + co);
#pragma unroll
            for (int j = 0; j < 3; ++j)
#pragma unroll
                for (int i = 0; i < 2; ++i) if (!(abl & 16)) acc[j][i] = __builtin_amdgcn_mfma_i32_32x32x32_i8(fw[j], fa[i], acc[j][i], 0, 0, 0);
        }
        if (kt + 1 < nkt && !(abl & 32)) stage(cur ^ 1);
        __syncthreads();
    }
    // ---- epilogue (the stage buffers are free now: red = [2][4 quarters][128 tokens] f32 at 0, a 12-KiB area per wave behind it: the residual tile,
    // then -- its first 4 KiB -- the output scratch)
    float *red = reinterpret_cast<float *>(smem);
    unsigned char *scr = smem + 4096 + wave * 12288;
    uint32_t klo = 0xFFFFFFFFu, khi = 0u;
    float v[2][3][16];
    float mean[2], inv[2];
    // The residual of this wave's two 32-token x 96-feature blocks: 2 x 12 coalesced requests (384-byte row segments, three whole cache lines each),
    // turned into the accumulator layout through the wave's LDS area (16-byte chunks XOR-swizzled by row & 7). Read straight in the accumulator layout
    // (round 3) an instruction touches 32 rows x 32 bytes and the address path, not HBM, sets the pace: the same change took 12 000 cycles per
    // sequence off attn_out_ln_quant_seq_kernel.
    f32x4q rres[2][3][4];
    {
        f32x4q ld[2][12];
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int r0 = m0 + wr * 64 + i * 32;
#pragma unroll
            for (int k = 0; k < 12; ++k) {
                const int p = k * 64 + lane, row = p / 24, cc = p % 24;
                int rg = r0 + row; if (rg >= M) rg = M - 1;
                if (!(abl & 4)) ld[i][k] = *reinterpret_cast<const f32x4q *>(reinterpret_cast<const unsigned char *>(resid + (size_t)rg * KT_NF + wc * 96) + cc * 16);
                else ld[i][k] = f32x4q{0.f, 0.f, 0.f, 0.f};
            }
        }
#pragma unroll
        for (int i = 0; i < 2; ++i) {
#pragma unroll
            for (int k = 0; k < 12; ++k) {
                const int p = k * 64 + lane, row = p / 24, cc = p % 24;
                *reinterpret_cast<f32x4q *>(scr + row * 384 + (((cc & ~7) | ((cc & 7) ^ (row & 7))) << 4)) = ld[i][k];
            }
#pragma unroll
            for (int j = 0; j < 3; ++j)
#pragma unroll
                for (int g = 0; g < 4; ++g) rres[i][j][g] = *reinterpret_cast<const f32x4q *>(scr + l31 * 384 + ((j * 8 + ((2 * g + hi) ^ (l31 & 7))) << 4));
        }
    }
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int rsa = ZWK ? rs_acc[i] + __shfl_xor(rs_acc[i], 32) : 0;
        float s = 0.0f;
#pragma unroll
        for (int j = 0; j < 3; ++j)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int nl = wc * 96 + j * 32 + 8 * g + 4 * hi;
                const f32x4q ws = *reinterpret_cast<const f32x4q *>(c_ws + nl), b4 = *reinterpret_cast<const f32x4q *>(c_b + nl);
                const i32x4q rz = *reinterpret_cast<const i32x4q *>(c_rz + nl), z4 = *reinterpret_cast<const i32x4q *>(c_zw + nl);
                const f32x4q r4 = rres[i][j][g];
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const float x = (float)(acc[j][i][4 * g + e] + __mul24(corr, rz[e]) - (ZWK ? __mul24(z4[e], rsa) : 0)) * (a_scale * ws[e]) + b4[e] + r4[e];
                    v[i][j][4 * g + e] = x; s += x;
                }
            }
        s += __shfl_xor(s, 32);
        if (hi == 0) red[wc * 128 + wr * 64 + i * 32 + l31] = s;
    }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int tl = wr * 64 + i * 32 + l31;
        mean[i] = (red[tl] + red[128 + tl] + red[256 + tl] + red[384 + tl]) * (1.0f / (float)KT_NF);
        float q = 0.0f;
#pragma unroll
        for (int j = 0; j < 3; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) { v[i][j][r] -= mean[i]; q += v[i][j][r] * v[i][j][r]; }
        q += __shfl_xor(q, 32);
        if (hi == 0) red[512 + wc * 128 + tl] = q;
    }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int tl = wr * 64 + i * 32 + l31;
        const float var = (red[512 + tl] + red[512 + 128 + tl] + red[512 + 256 + tl] + red[512 + 384 + tl]) * (1.0f / (float)KT_NF);
        inv[i] = 1.0f / sqrtf(var + eps);
    }
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const bool valid = m0 + wr * 64 + i * 32 + l31 < M;
#pragma unroll
        for (int j = 0; j < 3; ++j) {
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int nl = wc * 96 + j * 32 + 8 * g + 4 * hi;
                const f32x4q g4 = *reinterpret_cast<const f32x4q *>(c_g + nl), be4 = *reinterpret_cast<const f32x4q *>(c_be + nl);
                f32x4q ov;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    ov[e] = v[i][j][4 * g + e] * inv[i] * g4[e] + be4[e];
                    if (valid) { const uint32_t kk = order_key(ov[e]); klo = min(klo, kk); khi = max(khi, kk); }
                }
                *reinterpret_cast<f32x4q *>(scr + l31 * 128 + (((2 * g + hi) ^ (l31 & 7)) << 4)) = ov;
            }
#pragma unroll
            for (int h = 0; h < 4; ++h) {
                const int tl = h * 8 + (lane >> 3), ch = lane & 7;
                const f32x4q v4 = *reinterpret_cast<const f32x4q *>(scr + tl * 128 + ((ch ^ (tl & 7)) << 4));
                const int mt = m0 + wr * 64 + i * 32 + tl;
                if (mt < M && !(abl & 8)) *reinterpret_cast<f32x4q *>(out_f + (size_t)mt * KT_NF + wc * 96 + j * 32 + ch * 4) = v4;
            }
        }
    }
    if (mm_out) {
        for (int ofs = 32; ofs > 0; ofs >>= 1) { klo = min(klo, (uint32_t)__shfl_xor((int)klo, ofs)); khi = max(khi, (uint32_t)__shfl_xor((int)khi, ofs)); }
        if (lane == 0) {
            if (klo < __atomic_load_n(mm_out, __ATOMIC_RELAXED)) atomicMin(mm_out, klo);
            if (khi > __atomic_load_n(mm_out + 1, __ATOMIC_RELAXED)) atomicMax(mm_out + 1, khi);
        }
    }
}

// {scale, zp} of a tensor whose range keys are known (for the kernels that take the parameters instead of the keys)
__global__ void params_from_range_kernel(const uint32_t *__restrict__ mm, float *__restrict__ params) {
    if (threadIdx.x == 0 && blockIdx.x == 0) { const ActQ p = act_params(mm); params[0] = p.scale; params[1] = (float)p.zp; }
}

}  // namespace shodh
