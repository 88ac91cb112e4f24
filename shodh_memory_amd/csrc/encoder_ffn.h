// encoder_ffn.h -- the whole feed-forward half of a BERT layer in ONE kernel (bf16 mode; included by encoder.hip only):
//     X_next = LayerNorm( X + b2 + W2 * gelu(W1 * X + b1) )              (minilm.rs:939-949 runs this inside ONNX Runtime)
// Round 1 ran it as FFN-up (K = 384 streaming GEMM, 480 us per layer at 275k tokens), FFN-down (tiled K = 1536 GEMM, 584 us) and a
// LayerNorm kernel (110 us): the 1536-wide intermediate went out to HBM and back (845 MB each way per layer) and the f32
// pre-norm sums once more (422 MB each way). Here neither leaves the chip.
//
// Shape. Persistent workgroups of EIGHT waves, a tile = 128 tokens = 4 token blocks of 32. Token block tb is owned by a PAIR of
// waves that share a SIMD (waves tb and tb + 4):
//   producer (wave tb):     bx[24]  its 32 tokens x 384 as MFMA B fragments (lane = token), loaded once per tile          96 VGPRs
//                           per chunk c of 32 intermediate features (1536 = 48 chunks):  h(c) = W1[c] * X   (24 MFMAs, A = W1
//                           fragments from LDS), then hb(c) = bf16(gelu(h(c) + b1)) -> 2 KiB of LDS, ALREADY in B-fragment order
//   consumer (wave tb + 4): y[12]   the 32 x 384 output accumulators (lane = token, registers along the features)       192 VGPRs
//                           per chunk: y += W2[:, c] * hb(c)   (24 MFMAs, A = W2 fragments from LDS, B = the producer's 2 KiB)
// The C layout of the first product hands a lane 16 features of its token; the k-index of the second product is DEFINED as that
// order (W2 is packed to match at weight load), so hb needs no transposition -- the attention kernel's trick. A first version ran
// both products in ONE wave per SIMD (512 registers): correct, but one wave has ~5 issue slots per MFMA and the GELU alone
// needs 4 per MFMA: 3170 cycles per chunk against 1536 of matrix work. With two waves per SIMD the producer's VALU work
// issues while the consumer's MFMAs run.
// The weights stream through LDS by LDS-DMA in 24 KiB packages (one W1 chunk or one W2 chunk, both fragment-major: a wave's
// ds_read_b128 is one contiguous KiB, conflict-free without swizzling), five slots; every CU streams all of W1 and W2
// (2.25 MiB, L2-resident) once per 128-token tile: ~32 B/clk/CU.
// Pipeline iteration `it` of a tile (50 per tile: 48 chunks + 2 to drain), one s_barrier each:
//   producer: GEMM1(it) with gelu(h(it - 1)) in its shadows, hb(it - 1) -> LDS buffer (it - 1) & 1
//   consumer: GEMM2(it - 2) from LDS buffer it & 1
// The consumer's epilogue adds b2 and the residual, normalises the row (a lane holds 192 of its token's 384 features, the other
// half-wave the rest: one exchange) and stores bf16, while the producer already loads the next tile's tokens.
#pragma once
#include <type_traits>

#include "common.h"
#include "glds.h"

namespace shodh {

constexpr int FF_H = 384, FF_I = 1536, FF_KS = 24, FF_NB = 12, FF_CHUNKS = 48, FF_TOK = 128;
constexpr int FF_PKG = 24 * 1024;              // bytes of one package (W1 chunk: 24 k-steps x 1 KiB; W2 chunk: 12 n-blocks x 2 k-steps x 1 KiB)
constexpr int FF_SLOTS = 5;
constexpr int FF_HB = 2 * 4 * 2048;            // hb: 2 buffers x 4 token blocks x 2 KiB
constexpr int FF_LDS = FF_SLOTS * FF_PKG + FF_HB + FF_I * 4 + 5 * FF_H * 4;      // + b1, b2, gamma, beta, and the input LayerNorm's gamma, beta in LDS

// W2 [384][1536] row-major -> [chunk c][n-block nb][k-step ks][lane][8]: element = W2[32 nb + (lane & 31)][32 c + f(8 ks + j, lane >> 5)],
// f(r, hi) = (r & 3) + 8 (r >> 2) + 4 hi  -- the feature that register r of half-wave hi holds in the 32x32 C layout
__global__ void pack_w2_kernel(const __bf16 *__restrict__ W2, __bf16 *__restrict__ out) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (size_t)FF_H * FF_I) return;
    const int j = (int)(i & 7), lane = (int)((i >> 3) & 63), ks = (int)((i >> 9) & 1);
    const int nb = (int)((i >> 10) % FF_NB), c = (int)((i >> 10) / FF_NB);
    const int r = 8 * ks + j, hi = lane >> 5;
    const int f = (r & 3) + 8 * (r >> 2) + 4 * hi;
    out[i] = W2[(size_t)(nb * 32 + (lane & 31)) * FF_I + c * 32 + f];
}

typedef __bf16 bf16x8f __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4f __attribute__((ext_vector_type(4)));
typedef float f32x16f __attribute__((ext_vector_type(16)));
typedef float f32x4f __attribute__((ext_vector_type(4)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

// gelu(x) = x Phi(x) with Phi(x) = 0.5 + xc P(xc^2), xc = clamp(x, -4, 4), P of degree 7 (least-squares fit at Chebyshev nodes):
// |error| <= 3e-4 absolute over all x (bf16 rounds the result to 2^-9 relative anyway), 12 VALU operations and no transcendental.
// One wave per SIMD hides about six issue slots per MFMA: the erf form of the other kernels (~25 operations per value) made this
// kernel VALU-issue-bound at 3460 cycles per iteration against 1536 of MFMA work. (Tried: pairs of values on the packed FP32 pipe,
// v_pk_fma_f32 -- 916 us per layer against 824: the constants need register pairs and the packed form issues no faster here.)
__device__ __forceinline__ float ffn_gelu(float x) {
    const float xc = __builtin_fminf(__builtin_fmaxf(x, -4.0f), 4.0f);
    const float u = xc * xc;
    float p = __builtin_fmaf(-1.520480094e-09f, u, 1.180964698e-07f);
    p = __builtin_fmaf(p, u, -4.014221545e-06f);
    p = __builtin_fmaf(p, u, 7.960997465e-05f);
    p = __builtin_fmaf(p, u, -1.041295800e-03f);
    p = __builtin_fmaf(p, u, 9.641715296e-03f);
    p = __builtin_fmaf(p, u, -6.614117438e-02f);
    p = __builtin_fmaf(p, u, 3.988329119e-01f);
    return x * __builtin_fmaf(xc, p, 0.5f);
}

// ABL: diagnostics only (build with -DSHODH_FFN_ABLATE, pick with SHODH_FFN_ABLATE=<mask> at run time; results are INVALID for ABL != 0):
//   1 no GELU arithmetic   2 no epilogue (residual / LayerNorm / stores)   4 no GEMM1 MFMAs   8 no GEMM2 MFMAs   16 no weight DMA
//   32 no LDS reads of the weight fragments
template <int ABL>
__global__ __launch_bounds__(512, 2) void ffn_fused_kernel(const __bf16 *X /* may alias out: in place */, const __bf16 *__restrict__ W1p /* fragment-major FFN-up, chunk-major */,
                                                            const __bf16 *__restrict__ W2p /* pack_w2_kernel */, const float *__restrict__ b1,
                                                            const float *__restrict__ b2, const float *__restrict__ gamma, const float *__restrict__ beta,
                                                            const float *__restrict__ in_gamma /* LayerNorm applied to X on the way in, or null */,
                                                            const float *__restrict__ in_beta, __bf16 *out, int M, float eps) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const uint32_t smem_lds = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) unsigned char *)smem;
    unsigned char *hbs = smem + FF_SLOTS * FF_PKG;
    float *b1s = reinterpret_cast<float *>(hbs + FF_HB);
    const int tid = threadIdx.x, lane = tid & 63, hi = lane >> 5, l31 = lane & 31;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const bool producer = wave < 4;                      // wave-uniform
    const int tb = wave & 3;
    const int n_tiles = (M + FF_TOK - 1) / FF_TOK;
    float *b2s = b1s + FF_I, *gms = b2s + FF_H, *bts = gms + FF_H, *igs = bts + FF_H, *ibs = igs + FF_H;
    const bool in_ln = in_gamma != nullptr;              // uniform
    for (int i = tid; i < FF_I; i += 512) b1s[i] = b1[i];
    for (int i = tid; i < FF_H; i += 512) { b2s[i] = b2[i]; gms[i] = gamma[i]; bts[i] = beta[i]; igs[i] = in_ln ? in_gamma[i] : 1.0f; ibs[i] = in_ln ? in_beta[i] : 0.0f; }
    __syncthreads();                                     // the consumers read b2 before the first pipeline barrier

    const unsigned char *w1b = reinterpret_cast<const unsigned char *>(W1p), *w2b = reinterpret_cast<const unsigned char *>(W2p);
    // Package stream Q: Q(2 it) = W1 chunk it, Q(2 it + 1) = W2 chunk it - 2 (clamped to valid chunks at the ends: the piece count
    // per iteration stays constant, which is what the counted waits rely on); package s lives in slot s % 5. 24 pieces of 1 KiB per
    // package, 3 per wave.
    // The DMA is issued by the CONSUMER waves only, 6 pieces per package each, spread over their MFMA stream: an LDS-DMA
    // instruction stalls its wave 100-185 cycles at issue, and the consumer (24 MFMAs and 26 LDS reads per iteration) is the wave
    // with slack -- the producer on the same SIMD keeps the pipes busy meanwhile. (All eight waves issuing their share in a burst
    // at the top of the iteration cost ~1500 cycles per iteration.)
    const uint32_t lane16 = (uint32_t)lane * 16u;
    const int cw = wave & 3;                             // consumer index 0..3 (pieces cw*6 .. cw*6+5 of every package)
    struct Pkg { const unsigned char *src; uint32_t dst; };
    auto pkg_of = [&](uint32_t sq /* stream index */, int it_of /* iteration the package belongs to */) -> Pkg {
        const unsigned char *src;
        if ((sq & 1u) == 0u) { const int c1 = it_of < FF_CHUNKS ? it_of : FF_CHUNKS - 1; src = w1b + (size_t)c1 * FF_PKG; }
        else { const int c2 = it_of >= 2 ? (it_of - 2 < FF_CHUNKS ? it_of - 2 : FF_CHUNKS - 1) : 0; src = w2b + (size_t)c2 * FF_PKG; }
        Pkg p;
        p.src = uniform_ptr(src + (size_t)cw * 6 * 1024);
        p.dst = (uint32_t)__builtin_amdgcn_readfirstlane((int)(smem_lds + (sq % (uint32_t)FF_SLOTS) * FF_PKG + (uint32_t)cw * 6u * 1024u));
        return p;
    };
    auto issue_piece = [&](const Pkg &p, int i) { if (!(ABL & 16)) glds16(p.src, lane16 + i * 1024, p.dst + i * 1024); };
    constexpr int NIT = FF_CHUNKS + 2;
    uint32_t gq = 0;                                     // stream index of Q(2 it) of the current iteration; continues across tiles
    int tile = blockIdx.x;
    if (tile < n_tiles && !producer) {
        const Pkg p0 = pkg_of(0, 0), p1 = pkg_of(1, 0), p2 = pkg_of(2, 1);
#pragma unroll
        for (int i = 0; i < 6; ++i) issue_piece(p0, i);
#pragma unroll
        for (int i = 0; i < 6; ++i) issue_piece(p1, i);
#pragma unroll
        for (int i = 0; i < 6; ++i) issue_piece(p2, i);
    }

    const f32x16f zero16 = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
#ifdef SHODH_PROF
    long long pf_[10] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0}, tq_ = clock64();
#define FFP(i) { const long long t_ = clock64(); pf_[i] += t_ - tq_; tq_ = t_; }
#else
#define FFP(i)
#endif
    for (; tile < n_tiles; tile += gridDim.x) {
        const int tok = tile * FF_TOK + tb * 32 + l31;
        // Hand-over at the top of every iteration: the consumers wait until this iteration's two packages have landed (the 6 pieces of
        // the youngest package they issued may still fly), then the barrier tells everybody; it also means everybody is done with the
        // previous iteration's slots and hb buffer.
        if (producer) {
            // ---- producer: GEMM1 + GELU ---------------------------------------------------------------------------------------
            bf16x8f bx[FF_KS];
            {
                const bf16x8f *xp = reinterpret_cast<const bf16x8f *>(X + (size_t)tok * FF_H) + hi;      // rows past M: padding of the buffer, never stored
#pragma unroll
                for (int ks = 0; ks < FF_KS; ++ks) bx[ks] = xp[ks * 2];
            }
            if (in_ln) {
                // X holds the PRE-norm sum (attention output + residual): its LayerNorm happens here, on the fragments this lane holds
                // (192 of its token's 384 values; the other half-wave has the rest). The normalised rows are both this kernel's input
                // and its residual (handed to the consumer from these registers), and nobody else needs them.
                float s1 = 0.0f;
#pragma unroll
                for (int ks = 0; ks < FF_KS; ++ks)
#pragma unroll
                    for (int j = 0; j < 8; ++j) s1 += (float)bx[ks][j];
                s1 += __shfl_xor(s1, 32);
                const float mean = s1 * (1.0f / (float)FF_H);
                float vs = 0.0f;
#pragma unroll
                for (int ks = 0; ks < FF_KS; ++ks)
#pragma unroll
                    for (int j = 0; j < 8; ++j) { const float d = (float)bx[ks][j] - mean; vs += d * d; }
                vs += __shfl_xor(vs, 32);
                const float inv = 1.0f / __builtin_sqrtf(vs * (1.0f / (float)FF_H) + eps);
#pragma unroll
                for (int ks = 0; ks < FF_KS; ++ks) {
                    const f32x4f g0 = *reinterpret_cast<const f32x4f *>(igs + 16 * ks + 8 * hi), g1 = *reinterpret_cast<const f32x4f *>(igs + 16 * ks + 8 * hi + 4);
                    const f32x4f c0 = *reinterpret_cast<const f32x4f *>(ibs + 16 * ks + 8 * hi), c1 = *reinterpret_cast<const f32x4f *>(ibs + 16 * ks + 8 * hi + 4);
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        bx[ks][j] = (__bf16)(((float)bx[ks][j] - mean) * inv * g0[j] + c0[j]);
                        bx[ks][4 + j] = (__bf16)(((float)bx[ks][4 + j] - mean) * inv * g1[j] + c1[j]);
                    }
                }
            }
            f32x16f h_cur = zero16, h_odd = zero16, h_prev = zero16;
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_waitcnt(0x0F70);
            FFP(0)
            for (int it = 0; it < NIT; ++it, gq += 2) {
                FFP(1)
                __builtin_amdgcn_s_barrier();
                __builtin_amdgcn_sched_barrier(0);
                FFP(3)
                const bool g1 = it < FF_CHUNKS, ge = it >= 1 && it <= FF_CHUNKS;           // wave-uniform
                const bf16x8f *a1 = reinterpret_cast<const bf16x8f *>(smem + (gq % (uint32_t)FF_SLOTS) * FF_PKG) + lane;
                f32x4f bv[4];
#pragma unroll
                for (int g = 0; g < 4; ++g) bv[g] = *reinterpret_cast<const f32x4f *>(b1s + (ge ? it - 1 : 0) * 32 + 8 * g + 4 * hi);
                constexpr int D = (ABL & 128) ? 7 : 4;
                bf16x8f ring[8];
#pragma unroll
                for (int q = 0; q < D; ++q) ring[q] = a1[(ABL & 32) ? 0 : q * 64];
                bf16x8f hb[2];
#pragma unroll
                for (int st = 0; st < FF_KS; ++st) {
                    if (st + D < FF_KS && !(ABL & 32)) ring[(st + D) & 7] = a1[(st + D) * 64];
                    // two independent accumulator chains (even / odd k-steps): a chain of 24 dependent MFMAs waits for each predecessor
                    if (ABL & 4) { asm volatile("" : "+v"(ring[st & 7]), "+v"(bx[st])); }
                    else if ((st & 1) == 0) h_cur = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ring[st & 7], bx[st], st == 0 ? zero16 : h_cur, 0, 0, 0);
                    else h_odd = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ring[st & 7], bx[st], st == 1 ? zero16 : h_odd, 0, 0, 0);
                    if (st < 16) {      // gelu of h(it - 1) in the MFMA shadows; the empty asm pins the value to THIS step (the optimiser otherwise
                        float gv = (ABL & 1) ? h_prev[st] + bv[st >> 2][st & 3] : ffn_gelu(h_prev[st] + bv[st >> 2][st & 3]);      // sinks all sixteen evaluations below the loop, next to their only use)
                        asm volatile("" : "+v"(gv));
                        hb[st >> 3][st & 7] = (__bf16)gv;
                    }
                    __builtin_amdgcn_sched_barrier(0);
                }
                if (ge) {
                    // hb(it - 1) as the consumer's B fragments: [k-step][lane][8] (the lane's own registers, in order)
                    bf16x8f *dst = reinterpret_cast<bf16x8f *>(hbs + (((it - 1) & 1) * 4 + tb) * 2048) + lane;
                    dst[0] = hb[0]; dst[64] = hb[1];
                }
                if (g1) {
#pragma unroll
                    for (int r = 0; r < 16; ++r) h_prev[r] = h_cur[r] + h_odd[r];
                }
            }
            FFP(1)
            __builtin_amdgcn_s_barrier();          // pairs with the consumers' post-loop barrier (the last iteration's ring slots become their scratch)
            if (!(ABL & 2)) {
                // The residual IS this wave's bx: it goes to the paired consumer's scratch in two halves of 192 features (a lane's 16-B
                // fragment for k-step ks is chunk 2 ks + hi of its token's row) -- no second trip to HBM for it (the consumer used to
                // stage the rows itself, four exposed HBM round trips per tile with no registers to spare for more loads in flight).
                const uint32_t pfree0 = (gq + 3u) % (uint32_t)FF_SLOTS, pfree1 = (gq + 4u) % (uint32_t)FF_SLOTS;
                unsigned char *pscr = smem + (tb < 2 ? pfree0 : pfree1) * FF_PKG + (tb & 1) * 12288;
#pragma unroll
                for (int half = 0; half < 2; ++half) {
#pragma unroll
                    for (int j = 0; j < 12; ++j)
                        *reinterpret_cast<bf16x8f *>(pscr + l31 * 384 + (((2 * j + hi) ^ (l31 & 7)) << 4)) = bx[half * 12 + j];
                    __builtin_amdgcn_s_waitcnt(0xC07F);
                    __builtin_amdgcn_s_barrier();      // half written
                    if (half == 0) __builtin_amdgcn_s_barrier();      // half 0 consumed: the scratch can take half 1
                }
            }
        } else {
            // ---- consumer: GEMM2 + epilogue ------------------------------------------------------------------------------------
            // the accumulators start from b2 (lane = token, register r of block nb = feature 32 nb + (r & 3) + 8 (r >> 2) + 4 hi)
            f32x16f y[FF_NB];
#pragma unroll
            for (int nb = 0; nb < FF_NB; ++nb)
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const f32x4f c4 = *reinterpret_cast<const f32x4f *>(b2s + nb * 32 + 8 * g + 4 * hi);
#pragma unroll
                    for (int e = 0; e < 4; ++e) y[nb][4 * g + e] = c4[e];
                }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");       // the previous tile's stores
            __builtin_amdgcn_s_waitcnt(0x0F70);
            FFP(0)
            for (int it = 0; it < NIT; ++it, gq += 2) {
                FFP(1)
                asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
                FFP(2)
                __builtin_amdgcn_s_barrier();
                __builtin_amdgcn_sched_barrier(0);
                FFP(3)
                // Q(2 it + 3) belongs to iteration it + 1, Q(2 it + 4) to iteration it + 2 (of the next tile past the end); their slots
                // were the previous iteration's
                const int i1 = it + 1 < NIT ? it + 1 : it + 1 - NIT, i2 = it + 2 < NIT ? it + 2 : it + 2 - NIT;
                const Pkg pa = pkg_of(gq + 3, i1), pb = pkg_of(gq + 4, i2);
                if (it >= 2) {
                    const bf16x8f *a2 = reinterpret_cast<const bf16x8f *>(smem + ((gq + 1) % (uint32_t)FF_SLOTS) * FF_PKG) + lane;
                    const bf16x8f *hp = reinterpret_cast<const bf16x8f *>(hbs + ((it & 1) * 4 + tb) * 2048) + lane;
                    bf16x8f hb[2];
                    hb[0] = hp[0]; hb[1] = hp[64];
                    constexpr int D = 6;
                    bf16x8f ring[8];
#pragma unroll
                    for (int q = 0; q < D; ++q) ring[q] = a2[(ABL & 32) ? 0 : ((ABL & 64) ? ((q % FF_NB) * 2 + q / FF_NB) * 64 : q * 64)];
#pragma unroll
                    for (int st = 0; st < FF_KS; ++st) {       // st = 2 nb + ks
                        // (variant 64: all n-blocks of k-step 0, then of k-step 1 -- no MFMA depends on its predecessor)
                        auto frag = [&](int s_) { return (ABL & 64) ? ((s_ % FF_NB) * 2 + s_ / FF_NB) * 64 : s_ * 64; };
                        const int nb_ = (ABL & 64) ? st % FF_NB : st >> 1, ks_ = (ABL & 64) ? st / FF_NB : st & 1;
                        if (st + D < FF_KS && !(ABL & 32)) ring[(st + D) & 7] = a2[frag(st + D)];
                        if (ABL & 8) { asm volatile("" : "+v"(ring[st & 7]), "+v"(hb[ks_])); }
                        else y[nb_] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ring[st & 7], hb[ks_], y[nb_], 0, 0, 0);
                        if ((st & 1) == 1) { if (st < 12) issue_piece(pa, st >> 1); else issue_piece(pb, (st >> 1) - 6); }     // one piece every second MFMA
                        __builtin_amdgcn_sched_barrier(0);
                    }
                } else {
#pragma unroll
                    for (int i = 0; i < 6; ++i) issue_piece(pa, i);
#pragma unroll
                    for (int i = 0; i < 6; ++i) issue_piece(pb, i);
                }
            }
            FFP(1)
            // ---- epilogue: + b2 + residual, LayerNorm over the 384 features of each token, bf16 out -----------------------------
            // C layout: lane = token l31, register r of block nb = feature 32 nb + (r & 3) + 8 (r >> 2) + 4 hi: a lane's values are
            // 8-byte pieces scattered over 32 rows, and 48 such loads (residual) + 48 such stores per lane cost 84k cycles per tile
            // (store / load ISSUE bound: 32 partial lines per instruction). So both go through LDS: the two ring slots of the last
            // iteration are free after one more workgroup barrier (the DMA prefetch of the next tile runs in the other three), each
            // consumer wave gets 12 KiB = 32 rows x 192 features (half a row; two passes), rows in 16-B chunks XOR-swizzled by
            // (row & 7): global traffic is 16 B per lane along rows, the transposition happens in LDS.
            __builtin_amdgcn_s_barrier();
            FFP(5)
            if (ABL & 2) { asm volatile("" : "+v"(y[0]), "+v"(y[5]), "+v"(y[11])); continue; }
            const uint32_t free0 = (gq + 3u) % (uint32_t)FF_SLOTS, free1 = (gq + 4u) % (uint32_t)FF_SLOTS;     // = slots of Q(gq - 2), Q(gq - 1): the last iteration's
            unsigned char *scr = smem + (cw < 2 ? free0 : free1) * FF_PKG + (cw & 1) * 12288;
            const int tok0 = tile * FF_TOK + tb * 32;
            auto sw = [&](int row, int chunk) -> int { return row * 384 + ((chunk ^ (row & 7)) << 4); };
            // `opaque` makes the compiler recompute the (cheap) addresses at every step instead of hoisting ~100 of them into registers
            // next to the 192 accumulators (which spilled 349 registers to scratch and tripled the epilogue)
            auto opaque = [](int v) -> int { asm volatile("" : "+v"(v)); return v; };
            // the lane id from mbcnt (two VALU operations) rather than from a variable: with every register taken, `lane` itself lives
            // in scratch by now and each use would be a scratch reload (measured: 1200 cycles per epilogue step)
            auto lane_now = [&]() -> int { return opaque((int)__builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u))); };
            float s = 0.0f;
#pragma unroll
            for (int half = 0; half < 2; ++half) {
                // residual rows: written by the paired producer from its registers (see there)
                if (half == 1) __builtin_amdgcn_s_barrier();      // half 0 consumed
                __builtin_amdgcn_s_barrier();                      // this half written
                asm volatile("" ::: "memory"); __builtin_amdgcn_sched_barrier(0);
                FFP(6)
#pragma unroll
                for (int nbh = 0; nbh < 6; ++nbh)
#pragma unroll
                    for (int g = 0; g < 4; ++g) {
                        // one 8-byte LDS read per step and nothing hoisted: the 192 accumulators leave ~40 registers for everything else
                        const int nb = half * 6 + nbh, ln_ = lane_now(), lo = ln_ & 31, ho = ln_ >> 5;
                        const bf16x4f r4 = *reinterpret_cast<const bf16x4f *>(scr + sw(lo, nbh * 4 + g) + ho * 8);
#pragma unroll
                        for (int e = 0; e < 4; ++e) { const float v = y[nb][4 * g + e] + (float)r4[e]; y[nb][4 * g + e] = v; s += v; }
                        if (g == 3) { asm volatile("" ::: "memory"); __builtin_amdgcn_sched_barrier(0); }
                    }
            }
            FFP(7)
            s += __shfl_xor(s, 32);
            const float mean = s * (1.0f / (float)FF_H);
            float vs = 0.0f;
#pragma unroll
            for (int nb = 0; nb < FF_NB; ++nb)
#pragma unroll
                for (int r = 0; r < 16; ++r) { const float d = y[nb][r] - mean; vs += d * d; }
            vs += __shfl_xor(vs, 32);
            const float inv = 1.0f / __builtin_sqrtf(vs * (1.0f / (float)FF_H) + eps);
#pragma unroll
            for (int half = 0; half < 2; ++half) {
#pragma unroll
                for (int nbh = 0; nbh < 6; ++nbh)
#pragma unroll
                    for (int g = 0; g < 4; ++g) {
                        const int nb = half * 6 + nbh, ln_ = lane_now(), lo = ln_ & 31, ho = ln_ >> 5, n = nb * 32 + 8 * g + 4 * ho;
                        const f32x4f g4 = *reinterpret_cast<const f32x4f *>(gms + n), be4 = *reinterpret_cast<const f32x4f *>(bts + n);
                        bf16x4f o;
#pragma unroll
                        for (int e = 0; e < 4; ++e) {
                            // y a + b with a = inv gamma, b = beta - mean a: NOT (y - mean) ..., which the compiler shares with the variance
                            // pass above by parking all 192 differences in scratch (measured: 1200 cycles per step of reloads)
                            const float a_ = inv * g4[e];
                            o[e] = (__bf16)(y[nb][4 * g + e] * a_ + (be4[e] - mean * a_));
                        }
                        *reinterpret_cast<bf16x4f *>(scr + sw(lo, nbh * 4 + g) + ho * 8) = o;
                        if (g & 1) { asm volatile("" ::: "memory"); __builtin_amdgcn_sched_barrier(0); }      // two steps in flight
                    }
                FFP(8)
#pragma unroll
                for (int i = 0; i < 12; ++i) {
                    const int idx = i * 64 + lane_now(), row = idx / 24, ch = idx % 24;
                    const u32x4 v = *reinterpret_cast<const u32x4 *>(scr + sw(row, ch));
                    if (tok0 + row < M) *reinterpret_cast<u32x4 *>(out + (size_t)(tok0 + row) * FF_H + half * 192 + ch * 8) = v;
                    if ((i & 3) == 3) { asm volatile("" ::: "memory"); __builtin_amdgcn_sched_barrier(0); }
                }
            }
            FFP(4)
        }
#ifdef SHODH_PROF
        if (blockIdx.x == 100 && lane == 0 && tile + (int)gridDim.x >= n_tiles)
            printf("ffn wave %d (%s): tile-top %lld | work %lld | vmcnt-wait %lld | barrier %lld | epilogue(out rows) %lld | ep-barrier %lld resid-stage %lld passA %lld passB %lld\n", wave, producer ? "producer" : "consumer", pf_[0], pf_[1], pf_[2], pf_[3], pf_[4], pf_[5], pf_[6], pf_[7], pf_[8]);
#endif
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");     // the prefetch issued by the last iterations must not outlive the workgroup's LDS
}

}  // namespace shodh
