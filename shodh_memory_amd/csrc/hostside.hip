// hostside.hip -- the string / uuid glue on either side of the device path. These steps are O(k)
// or text hashing and stay on the host by design: the hash fallback embedder,
// search_ids post-processing and reciprocal-rank fusion. Plain C++; arithmetic order as the reference.
#include <algorithm>
#include <cmath>
#include <cstring>
#include <string>
#include <utility>
#include <vector>

#include "common.h"

#pragma clang fp contract(off)

namespace shodh {

// SipHash-1-3 with zero keys == std::collections::hash_map::DefaultHasher::new()
static inline uint64_t rotl(uint64_t x, int b) { return (x << b) | (x >> (64 - b)); }
static uint64_t siphash13(const uint8_t *in, size_t len) {
    uint64_t v0 = 0x736f6d6570736575ULL, v1 = 0x646f72616e646f6dULL, v2 = 0x6c7967656e657261ULL, v3 = 0x7465646279746573ULL;
    auto round = [&]() {
        v0 += v1; v1 = rotl(v1, 13); v1 ^= v0; v0 = rotl(v0, 32);
        v2 += v3; v3 = rotl(v3, 16); v3 ^= v2;
        v0 += v3; v3 = rotl(v3, 21); v3 ^= v0;
        v2 += v1; v1 = rotl(v1, 17); v1 ^= v2; v2 = rotl(v2, 32);
    };
    const size_t end = len - (len % 8);
    for (size_t i = 0; i < end; i += 8) {
        uint64_t m = 0;
        for (int j = 0; j < 8; ++j) m |= (uint64_t)in[i + j] << (8 * j);
        v3 ^= m; round(); v0 ^= m;
    }
    uint64_t b = (uint64_t)len << 56;
    for (size_t j = 0; j < len % 8; ++j) b |= (uint64_t)in[end + j] << (8 * j);
    v3 ^= b; round(); v0 ^= b;
    v2 ^= 0xff; round(); round(); round();
    return v0 ^ v1 ^ v2 ^ v3;
}
// impl Hash for str: the bytes followed by 0xFF
static uint64_t hash_str(const uint8_t *p, size_t n) {
    std::vector<uint8_t> buf(p, p + n);
    buf.push_back(0xff);
    return siphash13(buf.data(), buf.size());
}
static size_t u8len(uint8_t c) { return c < 0x80 ? 1 : (c >> 5) == 6 ? 2 : (c >> 4) == 14 ? 3 : (c >> 3) == 30 ? 4 : 1; }
static uint32_t u8dec(const uint8_t *p, size_t l) {
    if (l == 1) return p[0];
    if (l == 2) return ((p[0] & 0x1f) << 6) | (p[1] & 0x3f);
    if (l == 3) return ((p[0] & 0x0f) << 12) | ((p[1] & 0x3f) << 6) | (p[2] & 0x3f);
    return ((p[0] & 0x07) << 18) | ((p[1] & 0x3f) << 12) | ((p[2] & 0x3f) << 6) | (p[3] & 0x3f);
}
static bool is_ws(uint32_t c) {   // char::is_whitespace
    return (c >= 9 && c <= 13) || c == 0x20 || c == 0x85 || c == 0xA0 || c == 0x1680 || (c >= 0x2000 && c <= 0x200A) ||
           c == 0x2028 || c == 0x2029 || c == 0x202F || c == 0x205F || c == 0x3000;
}

struct UuidScore { uint8_t u[16]; float s; };
static bool uuid_score_less(const UuidScore &a, const UuidScore &b) {
    const uint32_t ka = order_key(a.s), kb = order_key(b.s);
    if (ka != kb) return ka > kb;                 // similarity / score descending (total_cmp)
    return memcmp(a.u, b.u, 16) < 0;              // MemoryId ascending (Uuid orders by its bytes)
}

}  // namespace shodh

using namespace shodh;

extern "C" {

int shodh_hash_embed(const char *utf8, size_t len, uint32_t dim, float *e) {      // minilm.rs:777-831
    if (!e || (len && !utf8) || dim == 0) { set_error("null argument"); return SHODH_ERR_INVALID; }
    const uint8_t *t = (const uint8_t *)utf8;
    for (uint32_t j = 0; j < dim; ++j) e[j] = 0.0f;
    size_t pos = 0, wi = 0;
    auto clen = [&](size_t p) { size_t l = u8len(t[p]); return p + l > len ? len - p : l; };
    while (pos < len) {
        while (pos < len) { const size_t l = clen(pos); if (!is_ws(u8dec(t + pos, l))) break; pos += l; }
        if (pos >= len) break;
        const size_t start = pos;
        while (pos < len) { const size_t l = clen(pos); if (is_ws(u8dec(t + pos, l))) break; pos += l; }
        const uint64_t h = hash_str(t + start, pos - start);
        for (size_t j = 0; j < dim; ++j) {
            const size_t index = (wi * 7 + j) % dim;
            const size_t bit = j < 64 ? j : (wi * 7 + j) % 64;
            e[index] = e[index] + (float)((h >> bit) & 1) * 0.1f;
        }
        ++wi;
    }
    size_t nchars = 0;
    for (size_t p = 0; p < len;) { p += clen(p); ++nchars; }
    if (nchars >= 2) {
        size_t p = 0;
        for (size_t i = 0; i + 1 < nchars; ++i) {
            const size_t l0 = clen(p), p1 = p + l0, l1 = clen(p1);
            const uint64_t h = hash_str(t + p, l0 + l1);
            for (size_t j = 0; j < 32; ++j) {
                const size_t index = (size_t)((h + (uint64_t)j) % (uint64_t)dim);
                e[index] = e[index] + (float)((h >> (j % 64)) & 1) * 0.05f;
            }
            p = p1;
        }
    }
    for (uint32_t j = 0; j < dim; ++j) if (std::isnan(e[j]) || std::isinf(e[j])) e[j] = 0.0f;
    float nsq = 0.0f;
    for (uint32_t j = 0; j < dim; ++j) nsq = nsq + e[j] * e[j];
    const float norm = std::sqrt(nsq);
    if (std::isnan(norm) || norm < 1.1920929e-07f) { for (uint32_t j = 0; j < dim; ++j) e[j] = 0.0f; return SHODH_OK; }
    for (uint32_t j = 0; j < dim; ++j) e[j] = e[j] / norm;
    return SHODH_OK;
}

size_t shodh_search_ids_postprocess(const uint32_t *vec_ids, const float *dists, size_t n_res, const uint8_t *v2m, size_t n_vectors,
                                    size_t limit, uint8_t *out_uuid, float *out_sim) {       // retrieval.rs:920-963
    static const uint8_t none[16] = {255, 255, 255, 255, 255, 255, 255, 255, 255, 255, 255, 255, 255, 255, 255, 255};
    std::vector<UuidScore> best;
    best.reserve(n_res);
    for (size_t r = 0; r < n_res; ++r) {
        const float similarity = -dists[r];
        if (vec_ids[r] >= n_vectors) continue;
        const uint8_t *u = v2m + (size_t)vec_ids[r] * 16;
        if (memcmp(u, none, 16) == 0) continue;
        size_t j = 0;
        for (; j < best.size(); ++j) if (memcmp(best[j].u, u, 16) == 0) break;
        if (j == best.size()) { UuidScore x; memcpy(x.u, u, 16); x.s = similarity; best.push_back(x); }
        else if (similarity > best[j].s) best[j].s = similarity;
    }
    std::sort(best.begin(), best.end(), uuid_score_less);
    const size_t out = best.size() < limit ? best.size() : limit;
    for (size_t i = 0; i < out; ++i) { memcpy(out_uuid + i * 16, best[i].u, 16); out_sim[i] = best[i].s; }
    return out;
}

size_t shodh_rrf_fuse(float k, const float *weights, size_t n_lists, const uint8_t *uuids, const size_t *list_len,
                      uint8_t *out_uuid, float *out_score, size_t out_cap) {                  // hybrid_search.rs:536-594
    std::vector<float> wn(n_lists ? n_lists : 1);
    float sum = 0.0f;
    for (size_t l = 0; l < n_lists; ++l) sum = sum + weights[l];
    for (size_t l = 0; l < n_lists; ++l) wn[l] = sum > 0.0f ? weights[l] / sum : 1.0f / (float)n_lists;
    std::vector<UuidScore> acc;
    size_t base = 0;
    for (size_t l = 0; l < n_lists; ++l) {
        for (size_t rank = 0; rank < list_len[l]; ++rank) {
            const uint8_t *u = uuids + (base + rank) * 16;
            const float contrib = wn[l] / (k + (float)(rank + 1));
            size_t j = 0;
            for (; j < acc.size(); ++j) if (memcmp(acc[j].u, u, 16) == 0) break;
            if (j == acc.size()) { UuidScore x; memcpy(x.u, u, 16); x.s = 0.0f; acc.push_back(x); }
            acc[j].s = acc[j].s + contrib;
        }
        base += list_len[l];
    }
    std::sort(acc.begin(), acc.end(), uuid_score_less);
    const size_t out = acc.size() < out_cap ? acc.size() : out_cap;
    for (size_t i = 0; i < out; ++i) { memcpy(out_uuid + i * 16, acc[i].u, 16); out_score[i] = acc[i].s; }
    return out;
}

// ---- ranking tail of RelevanceEngine::surface_relevant_inner (relevance.rs:801-918) --------------------------------
float shodh_calculate_tag_score(const char *context, const char *const *tags, size_t n_tags) {          // relevance.rs:680-705
    if (n_tags == 0 || !context || !tags) return 0.0f;
    auto lower = [](const char *z) { std::string o(z ? z : ""); for (char &c : o) if (c >= 'A' && c <= 'Z') c = (char)(c - 'A' + 'a'); return o; };
    const std::string ctx = lower(context);
    // split_whitespace: Unicode White_Space
    std::vector<std::pair<size_t, size_t>> words;
    {
        const uint8_t *t = (const uint8_t *)ctx.data();
        const size_t len = ctx.size();
        size_t pos = 0;
        auto clen = [&](size_t p) { size_t l = u8len(t[p]); return p + l > len ? len - p : l; };
        while (pos < len) {
            while (pos < len) { const size_t l = clen(pos); if (!is_ws(u8dec(t + pos, l))) break; pos += l; }
            if (pos >= len) break;
            const size_t st = pos;
            while (pos < len) { const size_t l = clen(pos); if (is_ws(u8dec(t + pos, l))) break; pos += l; }
            words.emplace_back(st, pos - st);
        }
    }
    size_t matches = 0;
    for (size_t i = 0; i < n_tags; ++i) {
        const std::string tag = lower(tags[i]);
        if (ctx.find(tag) != std::string::npos) { ++matches; continue; }
        for (const auto &wd : words) {
            const bool word_starts_with_tag = wd.second >= tag.size() && ctx.compare(wd.first, tag.size(), tag) == 0;
            const bool tag_starts_with_word = tag.size() >= wd.second && tag.compare(0, wd.second, ctx, wd.first, wd.second) == 0;
            if (word_starts_with_tag || tag_starts_with_word) { ++matches; break; }
        }
    }
    return (float)matches / (float)n_tags;
}

float shodh_apply_recency_boost(float base, int64_t age_hours_signed, uint64_t boost_hours, float multiplier) {   // relevance.rs:1524-1547
    if (boost_hours == 0) return base;
    const uint64_t age_hours = (uint64_t)age_hours_signed;
    if (age_hours > boost_hours) return base;
    const float decay = 1.0f - ((float)age_hours / (float)boost_hours);
    const float boost = 1.0f + (multiplier - 1.0f) * decay;
    const float v = base * boost;
    return v != v ? 1.0f : (v < 1.0f ? v : 1.0f);          // f32::min(1.0): a NaN operand is ignored
}

void shodh_relevance_cfg_default(shodh_relevance_cfg *c) {
    if (!c) return;
    c->min_importance = 0.3f; c->recency_boost_hours = 24; c->recency_boost_multiplier = 1.2f; c->graph_boost_multiplier = 1.15f; c->max_results = 5;
}

size_t shodh_rank_surfaced(const shodh_weights *w, const shodh_relevance_cfg *cfg, size_t n, const float *semantic, const float *entity,
                           const float *tag, const float *importance, const float *momentum_ema, const uint32_t *access_count,
                           const float *graph_strength, const int64_t *age_hours, const int64_t *created_at_ns, const uint8_t *uuid,
                           uint32_t *out_index, float *out_score, uint8_t *out_reason) {
    if (!w || !cfg || (n && (!semantic || !entity || !tag || !importance || !momentum_ema || !access_count || !graph_strength || !age_hours ||
                             !created_at_ns || !uuid))) { set_error("null argument"); return 0; }
    struct Surfaced { uint32_t index; float score; uint8_t reason; };
    std::vector<Surfaced> res;
    res.reserve(n);
    for (size_t i = 0; i < n; ++i) {
        if (importance[i] < cfg->min_importance) continue;                                                  // :809-812
        const float fused = shodh_fuse_scores_full(w, semantic[i], entity[i], tag[i], importance[i], momentum_ema[i], access_count[i], graph_strength[i]);
        const uint8_t reason = (semantic[i] > 0.0f && entity[i] > 0.0f) ? SHODH_REASON_COMBINED : entity[i] > 0.0f ? SHODH_REASON_ENTITY_MATCH
                             : semantic[i] > 0.0f ? SHODH_REASON_SEMANTIC_SIMILARITY : SHODH_REASON_RECENT_IMPORTANT;   // :858-866
        const float boosted = shodh_apply_recency_boost(fused, age_hours[i], cfg->recency_boost_hours, cfg->recency_boost_multiplier);
        float final_score = boosted;
        if (entity[i] > 0.0f) { const float g = boosted * cfg->graph_boost_multiplier; final_score = g != g ? 1.0f : (g < 1.0f ? g : 1.0f); }   // :877-881
        res.push_back(Surfaced{(uint32_t)i, final_score, reason});
    }
    std::stable_sort(res.begin(), res.end(), [&](const Surfaced &a, const Surfaced &b) {                    // :904-909
        const uint32_t ka = order_key(a.score), kb = order_key(b.score);
        if (ka != kb) return ka > kb;
        if (created_at_ns[a.index] != created_at_ns[b.index]) return created_at_ns[a.index] > created_at_ns[b.index];
        return memcmp(uuid + (size_t)a.index * 16, uuid + (size_t)b.index * 16, 16) < 0;
    });
    size_t out = 0;
    for (const Surfaced &r : res) {
        if (!(r.score >= 0.25f)) continue;                                                                   // MIN_RELEVANCE_SCORE (:913-914)
        if (out >= cfg->max_results) break;                                                                  // :917
        if (out_index) out_index[out] = r.index;
        if (out_score) out_score[out] = r.score;
        if (out_reason) out_reason[out] = r.reason;
        ++out;
    }
    return out;
}

}  // extern "C"
