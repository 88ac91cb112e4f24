// legfusion.hip -- recall "Layer 4": the hybrid (vector + BM25) leg and the graph leg fused into one score per memory
// (MemorySystem::recall, src/memory/mod.rs:3878-4468). This is the step that consumes search_ids' output, a few hundred
// candidates per request at most, so it is host C++. The reference reads its experiment switches from
// the environment in the middle of recall; here they are the fields of shodh_leg_fusion_cfg. Arithmetic order as the
// reference (f32 throughout, no contraction).
#include <algorithm>
#include <cmath>
#include <cstring>
#include <vector>

#include "common.h"

#pragma clang fp contract(off)

namespace shodh {
namespace {

// f32::clamp (a NaN stays a NaN) and f32::max (a NaN operand is ignored)
inline float clampf(float x, float lo, float hi) { return x < lo ? lo : (x > hi ? hi : x); }
inline float maxf(float a, float b) { return a != a ? b : (b != b ? a : (a > b ? a : b)); }

struct Component { uint8_t u[16]; float bm25, vec; };       // hybrid_components: id -> (bm25_score, vector_score)
struct Ranked { const uint8_t *u; float s; };
bool ranked_before(const Ranked &a, const Ranked &b) {     // b.1.total_cmp(&a.1).then_with(|| a.0.cmp(b.0))
    const uint32_t ka = order_key(a.s), kb = order_key(b.s);
    if (ka != kb) return ka > kb;
    return memcmp(a.u, b.u, 16) < 0;
}
struct Fused { uint8_t u[16]; float s; };
bool fused_before(const Fused &a, const Fused &b) {
    const uint32_t ka = order_key(a.s), kb = order_key(b.s);
    if (ka != kb) return ka > kb;
    return memcmp(a.u, b.u, 16) < 0;
}

// the per-leg score lists of the gate features: positive scores only, best first (mod.rs:4110-4122)
void positive_sorted(const std::vector<Component> &comp, bool vector_leg, std::vector<Ranked> &out) {
    out.clear();
    for (const Component &c : comp) {
        const float v = vector_leg ? c.vec : c.bm25;
        if (v > 0.0f) out.push_back(Ranked{c.u, v});
    }
    std::sort(out.begin(), out.end(), ranked_before);
}
float peakedness(const std::vector<Ranked> &xs) {          // max / mean, 1.0 for an empty or flat-zero leg (mod.rs:4123-4134)
    if (xs.empty()) return 1.0f;
    float sum = 0.0f;
    for (const Ranked &r : xs) sum = sum + r.s;
    const float mean = sum / (float)xs.size();
    return mean > 1e-6f ? xs[0].s / mean : 1.0f;
}
float top_overlap(const std::vector<Ranked> &by_vec, const std::vector<Ranked> &by_bm, size_t want) {   // mod.rs:4135-4147, :4191-4199
    size_t k = want;
    if (by_vec.size() < k) k = by_vec.size();
    if (by_bm.size() < k) k = by_bm.size();
    if (k < 1) k = 1;
    size_t hits = 0;
    for (size_t i = 0; i < k && i < by_bm.size(); ++i)
        for (size_t j = 0; j < k && j < by_vec.size(); ++j)
            if (memcmp(by_bm[i].u, by_vec[j].u, 16) == 0) { ++hits; break; }
    return (float)hits / (float)k;
}

// offline-fitted logistic gate: (mean, sd, weight) per standardised feature and the bias (mod.rs:4096-4109)
const float kFit[11][3] = {
    {2.77242f, 1.87083f, -0.301375f},    {1.3841f, 0.180661f, -0.0212517f},  {0.307389f, 0.170755f, -0.243719f},
    {93.4129f, 43.8602f, -0.556471f},    {0.597371f, 0.0823258f, 0.236463f}, {106.897f, 36.4931f, 0.597537f},
    {31.4138f, 7.54534f, -0.881755f},    {116.222f, 38.3149f, 0.582615f},    {9.90148f, 0.987682f, 0.304049f},
    {0.358194f, 0.0710801f, 0.0264384f}, {54.1034f, 16.0475f, -0.571797f},
};
const float kFitBias = -0.985886f;

}  // namespace
}  // namespace shodh

using namespace shodh;

extern "C" {

void shodh_leg_fusion_cfg_default(shodh_leg_fusion_cfg *c) {
    if (!c) return;
    memset(c, 0, sizeof(*c));
    c->flat_adaptive = 1; c->adapt_symmetric = 1;
    c->graph_w = 0.3f; c->hybrid_w = 0.6f + 0.1f;
    c->rrf_k = 30.0f; c->flat_consensus = 0.3f; c->adapt_trust_max = 2.0f;
    c->fw_graph = 0.3f; c->fw_vec = 0.6f; c->fw_bm25 = 0.4f;
    c->agree_k = 10.0f; c->agree_lo = 0.1f; c->agree_hi = 0.5f;
    c->peak_lo = 2.0f; c->peak_hi = 6.0f;
}

void shodh_density_weights(float d, float *out) {          // graph_retrieval.rs:81-101, constants.rs:478-510
    const float w_min = 0.1f, w_max = 0.5f, linguistic = 0.15f, d_lo = 0.5f, d_hi = 2.0f;
    float g;
    if (d <= d_lo) g = w_max;
    else if (d >= d_hi) g = w_min;
    else { const float ratio = (d - d_lo) / (d_hi - d_lo); g = w_max - ratio * (w_max - w_min); }
    out[0] = 1.0f - g - linguistic; out[1] = g; out[2] = linguistic;
}

void shodh_leg_fusion_weights(int has_density, float density, float override_w, float floor_w, float *graph_w, float *hybrid_w) {
    float w[3] = {0.6f, 0.3f, 0.1f};                         // mod.rs:3878-3880
    if (has_density) shodh_density_weights(density, w);
    float semantic = w[0], graph = w[1];
    const float linguistic = w[2];
    if (override_w == override_w) graph = clampf(override_w, 0.0f, 1.0f);        // SHODH_GRAPH_FUSION_WEIGHT (:3888-3892)
    if (floor_w == floor_w && floor_w > graph) {                                  // SHODH_GRAPH_W_FLOOR (:3902-3918)
        const float max_floor = maxf(1.0f - linguistic - 0.05f, 0.0f);
        const float requested = clampf(floor_w, 0.0f, 0.95f);
        graph = requested < max_floor ? requested : max_floor;                    // f32::min
        semantic = maxf(1.0f - graph - linguistic, 0.0f);
    }
    if (graph_w) *graph_w = graph;
    if (hybrid_w) *hybrid_w = semantic + linguistic;
}

size_t shodh_fuse_legs(const shodh_leg_fusion_cfg *c, const uint8_t *h_uuid, const float *h_bm25, const float *h_vec, size_t n_hybrid,
                       const uint8_t *g_uuid, const float *g_act, size_t n_graph, size_t query_len, uint8_t *out_uuid, float *out_score,
                       size_t out_cap, float *vec_trust_out) {
    if (!c || (n_hybrid && (!h_uuid || !h_bm25 || !h_vec)) || (n_graph && (!g_uuid || !g_act))) { set_error("null argument"); return 0; }
    // hybrid_components, in first-insertion order; a repeated id keeps its key and takes the later values (:3832-3838)
    std::vector<Component> comp;
    comp.reserve(n_hybrid);
    for (size_t i = 0; i < n_hybrid; ++i) {
        const uint8_t *u = h_uuid + i * 16;
        size_t j = 0;
        for (; j < comp.size(); ++j) if (memcmp(comp[j].u, u, 16) == 0) break;
        if (j == comp.size()) { Component x; memcpy(x.u, u, 16); comp.push_back(x); }
        comp[j].bm25 = h_bm25[i]; comp[j].vec = h_vec[i];
    }
    float max_activation = 0.0f;                              // :3938-3942
    for (size_t i = 0; i < n_graph; ++i) max_activation = maxf(max_activation, g_act[i]);
    const float graph_max_act = max_activation;               // the gate's feature is the un-floored maximum (:4148-4151)
    max_activation = maxf(max_activation, 1e-6f);
    const float n_graph_f = (float)(n_graph > 1 ? n_graph : 1), n_hybrid_f = (float)(n_hybrid > 1 ? n_hybrid : 1);

    // SHODH_LEG: keep one leg's candidates only (:3975-3988)
    if (c->isolate_leg == 1) {
        comp.erase(std::remove_if(comp.begin(), comp.end(), [](const Component &x) { return !(x.vec > 0.0f); }), comp.end());
        for (Component &x : comp) x.bm25 = 0.0f;
    } else if (c->isolate_leg == 2) {
        comp.erase(std::remove_if(comp.begin(), comp.end(), [](const Component &x) { return !(x.bm25 > 0.0f); }), comp.end());
        for (Component &x : comp) x.vec = 0.0f;
    } else if (c->isolate_leg == 3) {
        comp.clear();
    }
    const bool graph_leg_on = !(c->isolate_leg == 1 || c->isolate_leg == 2);
    const float flat_consensus = clampf(c->flat_consensus, 0.0f, 1.0f);
    float max_vec = 0.0f, max_bm = 0.0f;                      // :4003-4012
    for (const Component &x : comp) { max_vec = maxf(max_vec, x.vec); max_bm = maxf(max_bm, x.bm25); }
    max_vec = maxf(max_vec, 1e-6f); max_bm = maxf(max_bm, 1e-6f);
    const bool sum_fusion = c->fusion_sum != 0, v2_fusion = c->fusion_v2 != 0;
    const bool flat_fusion = c->fusion_flat || (!c->fusion_rrf && !v2_fusion && !sum_fusion);   // :4044

    // per-query trust in the vector leg (:4066-4251)
    float vec_trust = 1.0f;
    if (c->flat_adaptive) {
        float t;
        std::vector<Ranked> by_vec, by_bm;
        if (c->adapt_feature == 0) {                          // fitted logistic gate over eleven pool features
            positive_sorted(comp, true, by_vec);
            positive_sorted(comp, false, by_bm);
            const float agreement = (by_vec.empty() || by_bm.empty()) ? 0.0f : top_overlap(by_vec, by_bm, 10);
            const float feats[11] = {peakedness(by_bm), peakedness(by_vec), agreement, max_bm, max_vec, (float)by_bm.size(),
                                     (float)by_vec.size(), (float)comp.size(), (float)n_graph, graph_max_act, (float)query_len};
            float s = kFitBias;
            for (int i = 0; i < 11; ++i) s = s + kFit[i][2] * (feats[i] - kFit[i][0]) / kFit[i][1];
            t = 1.0f / (1.0f + std::exp(-clampf(s, -30.0f, 30.0f)));
        } else if (c->adapt_feature == 1) {                   // agreement of the two legs' top-K (:4169-4205)
            positive_sorted(comp, true, by_vec);
            positive_sorted(comp, false, by_bm);
            if (by_bm.empty()) t = 1.0f;
            else if (by_vec.empty()) t = 0.0f;
            else {
                const float kf = maxf(c->agree_k, 1.0f);      // `as usize` saturates
                const size_t want = kf >= 1.8446744e19f ? (size_t)-1 : (size_t)kf;
                const float overlap = top_overlap(by_vec, by_bm, want);
                const float span = maxf(c->agree_hi - c->agree_lo, 1e-6f);
                t = clampf((c->agree_hi - overlap) / span, 0.0f, 1.0f);
            }
        } else {                                              // BM25 peakedness (:4206-4228); the reference sums the positive scores
            float sum = 0.0f;                                 // in HashMap order, here in first-insertion order
            size_t cnt = 0;
            for (const Component &x : comp) if (x.bm25 > 0.0f) { sum = sum + x.bm25; ++cnt; }
            const float mean_bm = cnt ? sum / (float)cnt : 0.0f;
            const float bm_peak = mean_bm > 1e-6f ? max_bm / mean_bm : 1.0f;
            const float span = maxf(c->peak_hi - c->peak_lo, 1e-6f);
            t = clampf((c->peak_hi - bm_peak) / span, 0.0f, 1.0f);
        }
        vec_trust = c->adapt_symmetric ? maxf(1.0f + (c->adapt_trust_max - 1.0f) * (2.0f * t - 1.0f), 0.2f)     // :4245-4249
                                       : 1.0f + (c->adapt_trust_max - 1.0f) * t;
    }
    if (vec_trust_out) *vec_trust_out = vec_trust;

    std::vector<Fused> fused;
    fused.reserve(n_hybrid + n_graph);
    auto entry = [&](const uint8_t *u) -> float & {           // fused.entry(id).or_insert(0.0)
        for (Fused &f : fused) if (memcmp(f.u, u, 16) == 0) return f.s;
        Fused f; memcpy(f.u, u, 16); f.s = 0.0f; fused.push_back(f);
        return fused.back().s;
    };
    const float graph_w = c->graph_w, hybrid_w = c->hybrid_w, k = c->rrf_k;
    // graph leg (:4348-4413)
    for (size_t r = 0; r < n_graph && graph_leg_on; ++r) {
        const float activation = g_act[r];
        float score;
        if (sum_fusion) score = c->fw_graph * clampf(activation / max_activation, 0.0f, 1.0f);
        else if (v2_fusion) {
            const float borda = graph_w * ((n_graph_f - (float)r) / n_graph_f);
            const float rescue = graph_w * clampf(activation / max_activation, 0.0f, 1.0f);
            score = borda + rescue;
        } else if (flat_fusion) score = graph_w * clampf(activation / max_activation, 0.0f, 1.0f);
        else score = graph_w / (k + (float)(r + 1));
        float &slot = entry(g_uuid + r * 16);
        slot = slot + score;
        const float activation_factor = (v2_fusion || sum_fusion) ? 1.0f : 1.0f + graph_w * 0.3f * clampf(activation, 0.0f, 1.0f);   // ACTIVATION_BONUS_SCALE
        slot = slot * activation_factor;
    }
    // hybrid leg (:4414-4468)
    for (size_t r = 0; r < n_hybrid; ++r) {
        const uint8_t *u = h_uuid + r * 16;
        float bm25 = 0.0f, vec = 0.0f;                        // hybrid_components.get(id).unwrap_or((0.0, 0.0))
        for (const Component &x : comp) if (memcmp(x.u, u, 16) == 0) { bm25 = x.bm25; vec = x.vec; break; }
        float score;
        if (sum_fusion) score = c->fw_vec * clampf(vec / max_vec, 0.0f, 1.0f) + c->fw_bm25 * clampf(bm25 / max_bm, 0.0f, 1.0f);
        else if (flat_fusion) {
            const float vn = clampf(vec / max_vec, 0.0f, 1.0f) * vec_trust;
            const float bn = clampf(bm25 / max_bm, 0.0f, 1.0f);
            const float hi = vn >= bn ? vn : bn, lo = vn >= bn ? bn : vn;
            score = hybrid_w * (hi + flat_consensus * lo);
        } else if (v2_fusion) score = hybrid_w * ((n_hybrid_f - (float)r) / n_hybrid_f);
        else score = hybrid_w / (k + (float)(r + 1));
        float &slot = entry(u);
        slot = slot + score;
    }
    std::sort(fused.begin(), fused.end(), fused_before);
    const size_t out = fused.size() < out_cap ? fused.size() : out_cap;
    for (size_t i = 0; i < out; ++i) { if (out_uuid) memcpy(out_uuid + i * 16, fused[i].u, 16); if (out_score) out_score[i] = fused[i].s; }
    return fused.size();
}

}  // extern "C"
