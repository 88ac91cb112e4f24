// fusion.hip -- LearnedWeights score fusion (src/relevance.rs:343-606).
// Scalar f32 arithmetic in the reference's written order. The per-request candidate count is
// tens (relevance.rs:801-918), so the scalar entry points run on the host; the batch entry point
// evaluates large candidate lists on the device with the same expression tree.
#include <cmath>

#include "common.h"

#pragma clang fp contract(off)

namespace shodh {

__host__ __device__ static inline float calibrate(float s) {     // relevance.rs:601-606
    if (!(fabsf(s) <= 3.4028235e38f)) return 0.0f;                // !is_finite
    return 1.0f / (1.0f + expf(-10.0f * (s - 0.5f)));
}

__host__ __device__ static inline float fuse_full(const shodh_weights &w, float sem, float ent, float tag, float imp,
                                                  float mom, uint32_t acc, float gs) {   // relevance.rs:529-594
    const float c_sem = calibrate(sem), c_ent = calibrate(ent), c_tag = calibrate(tag), c_imp = calibrate(imp);
    const float nm = (mom + 1.0f) / 2.0f;
    float am;
    if (nm > 0.65f) am = fminf(nm * 1.5f, 1.0f);
    else if (nm < 0.40f) am = fmaxf(nm * 0.3f, 0.0f);
    else am = nm;
    const float c_mom = calibrate(am);
    float as = 0.0f;
    if (acc != 0) { const float la = log2f((float)acc + 1.0f); as = fminf(la / 4.0f, 1.0f); }
    const float c_acc = calibrate(as);
    const float c_gs = calibrate(gs);
    float r = w.semantic * c_sem + w.entity * c_ent;
    r = r + w.tag * c_tag;
    r = r + w.importance * c_imp;
    r = r + w.momentum * c_mom;
    r = r + w.access_count * c_acc;
    r = r + w.graph_strength * c_gs;
    return (fabsf(r) <= 3.4028235e38f) ? r : 0.0f;
}

__global__ void fuse_batch_kernel(shodh_weights w, uint64_t n, const float *sem, const float *ent, const float *tag,
                                  const float *imp, const float *mom, const uint32_t *acc, const float *gs, float *out) {
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = fuse_full(w, sem[i], ent[i], tag[i], imp[i], mom[i], acc[i], gs[i]);
}

}  // namespace shodh

using namespace shodh;

extern "C" {

void shodh_weights_default(shodh_weights *w) {        // relevance.rs:64-100, :383-397
    if (!w) return;
    w->semantic = 0.18f; w->entity = 0.17f; w->tag = 0.05f; w->importance = 0.05f;
    w->momentum = 0.28f; w->access_count = 0.14f; w->graph_strength = 0.13f; w->update_count = 0;
}

void shodh_weights_normalize(shodh_weights *w) {      // relevance.rs:401-418
    if (!w) return;
    float sum = w->semantic + w->entity;
    sum = sum + w->tag; sum = sum + w->importance; sum = sum + w->momentum;
    sum = sum + w->access_count; sum = sum + w->graph_strength;
    if (sum > 0.0f) {
        w->semantic /= sum; w->entity /= sum; w->tag /= sum; w->importance /= sum;
        w->momentum /= sum; w->access_count /= sum; w->graph_strength /= sum;
    }
}

void shodh_weights_apply_feedback(shodh_weights *w, int sem, int ent, int tag, int helpful) {   // relevance.rs:427-465
    if (!w) return;
    const float LR = 0.05f, MINW = 0.05f;
    const float direction = helpful ? 1.0f : -1.0f;
    const float delta = LR * direction;
    if (sem) w->semantic = fmaxf(w->semantic + delta, MINW);
    if (ent) w->entity = fmaxf(w->entity + delta, MINW);
    if (tag) w->tag = fmaxf(w->tag + delta, MINW);
    if (helpful && !sem && !ent && !tag) w->importance = fmaxf(w->importance + delta, MINW);
    const float aux = LR * direction * 0.5f;
    w->momentum = fmaxf(w->momentum + aux, MINW);
    w->access_count = fmaxf(w->access_count + aux, MINW);
    w->graph_strength = fmaxf(w->graph_strength + aux, MINW);
    shodh_weights_normalize(w);
    w->update_count += 1;
}

float shodh_calibrate_score(float score) { return calibrate(score); }

float shodh_fuse_scores_full(const shodh_weights *w, float sem, float ent, float tag, float imp, float mom,
                             uint32_t access_count, float graph_strength) {
    if (!w) return 0.0f;
    return fuse_full(*w, sem, ent, tag, imp, mom, access_count, graph_strength);
}
float shodh_fuse_scores(const shodh_weights *w, float sem, float ent, float tag, float imp) {       // relevance.rs:471-487
    return shodh_fuse_scores_full(w, sem, ent, tag, imp, 0.0f, 0, 0.5f);
}
float shodh_fuse_scores_with_momentum(const shodh_weights *w, float sem, float ent, float tag, float imp, float mom) {   // :499-517
    return shodh_fuse_scores_full(w, sem, ent, tag, imp, mom, 0, 0.5f);
}

int shodh_fuse_scores_full_batch(int device, const shodh_weights *w, uint64_t n, const float *sem, const float *ent,
                                 const float *tag, const float *imp, const float *mom, const uint32_t *acc,
                                 const float *gs, float *out) {
    if (!w || (n && (!sem || !ent || !tag || !imp || !mom || !acc || !gs || !out))) { set_error("null argument"); return SHODH_ERR_INVALID; }
    if (n == 0) return SHODH_OK;
    SHODH_HIP_TRY(hipSetDevice(device));
    float *d = nullptr;
    SHODH_HIP_TRY(dev_alloc((void **)&d, n * 4 * 8));
    const float *src[6] = {sem, ent, tag, imp, mom, gs};
    for (int i = 0; i < 6; ++i) SHODH_HIP_TRY(hipMemcpy(d + (size_t)i * n, src[i], n * 4, hipMemcpyHostToDevice));
    SHODH_HIP_TRY(hipMemcpy(d + 6 * n, acc, n * 4, hipMemcpyHostToDevice));
    hipLaunchKernelGGL(fuse_batch_kernel, dim3((uint32_t)ceil_div(n, 256)), dim3(256), 0, nullptr, *w, n, d, d + n, d + 2 * n,
                       d + 3 * n, d + 4 * n, (const uint32_t *)(d + 6 * n), d + 5 * n, d + 7 * n);
    hipError_t e = hipGetLastError();
    if (e == hipSuccess) e = hipMemcpy(out, d + 7 * n, n * 4, hipMemcpyDeviceToHost);
    dev_free(d);
    if (e != hipSuccess) { set_error("fuse batch failed: %s", hipGetErrorString(e)); return SHODH_ERR_DEVICE; }
    return SHODH_OK;
}

}  // extern "C"
