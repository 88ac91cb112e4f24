// persist.hip -- readers / writers of the reference's on-disk index formats (host code).
//   VAMA v1  src/vector_db/vamana_persist.rs:6-34 (layout), :46-49 (constants), :98-112 (header bytes), :155-163 (checksum),
//            :175-284 (save), :290-391 (load)
//   SPAN v1  src/vector_db/spann.rs:13-52 (layout), :76-80 (constants), :221-252 (header bytes), :750-876 (save), :879-1003 (load),
//            :1091-1098 (checksum)
// Both are little-endian, packed, and protected by FNV-1a-64 over every byte after the header. A persisted index of a real
// shodh data directory can be loaded straight into the GPU index (vectors + tombstones; centroids + codebook + postings).
// The Vamana graph section is carried through untouched (degree / neighbour arrays) for callers that want to write it
// back; this library itself searches exactly and writes empty adjacency for indexes it created.
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <vector>

#include "../../include/shodh_hip.h"
#include "common.h"

namespace shodh {
namespace {

constexpr size_t VAMA_HEADER = 64, SPAN_HEADER = 128, ALIGN = 64;

uint64_t fnv1a64(const unsigned char *p, size_t n) {
    uint64_t h = 0xcbf29ce484222325ull;
    for (size_t i = 0; i < n; ++i) { h ^= p[i]; h *= 0x100000001b3ull; }
    return h;
}
size_t align_to(size_t o, size_t a) { return (o + a - 1) & ~(a - 1); }
template <class T> T rd(const unsigned char *p) { T v; memcpy(&v, p, sizeof(T)); return v; }
template <class T> void wr(unsigned char *p, T v) { memcpy(p, &v, sizeof(T)); }

int read_file(const char *path, std::vector<unsigned char> &buf) {
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
int write_file(const char *path, const std::vector<unsigned char> &buf) {
    FILE *f = fopen(path, "wb");
    if (!f) { set_error("cannot create %s", path); return SHODH_ERR_IO; }
    const size_t put = buf.empty() ? 0 : fwrite(buf.data(), 1, buf.size(), f);
    const int rc = fclose(f);
    if (put != buf.size() || rc != 0) { set_error("short write on %s", path); return SHODH_ERR_IO; }
    return SHODH_OK;
}

// ---- VAMA ----
int vama_parse(const std::vector<unsigned char> &b, shodh_vama_info *o, size_t *graph_off, size_t *vec_off) {
    if (b.size() < VAMA_HEADER) { set_error("VAMA: header too small"); return SHODH_ERR_IO; }
    if (memcmp(b.data(), "VAMA", 4) != 0) { set_error("VAMA: invalid magic bytes"); return SHODH_ERR_IO; }
    if (rd<uint32_t>(&b[4]) != 1) { set_error("VAMA: unsupported version %u", rd<uint32_t>(&b[4])); return SHODH_ERR_UNSUPPORTED; }
    o->num_vectors = rd<uint64_t>(&b[8]);
    o->dimension = rd<uint32_t>(&b[16]);
    o->max_degree = rd<uint32_t>(&b[20]);
    o->medoid = rd<uint32_t>(&b[24]);
    o->distance_metric = b[28];
    o->deleted_count = rd<uint32_t>(&b[29]);
    o->incremental_inserts = rd<uint64_t>(&b[33]);
    const uint64_t stored = rd<uint64_t>(&b[41]);
    if (stored != fnv1a64(b.data() + VAMA_HEADER, b.size() - VAMA_HEADER)) { set_error("VAMA: checksum mismatch"); return SHODH_ERR_IO; }
    size_t off = VAMA_HEADER + (size_t)o->deleted_count * 4;
    if (off > b.size()) { set_error("VAMA: truncated deleted-id section"); return SHODH_ERR_IO; }
    *graph_off = off;
    uint64_t edges = 0;
    for (uint64_t i = 0; i < o->num_vectors; ++i) {
        if (off + 2 > b.size()) { set_error("VAMA: truncated graph section"); return SHODH_ERR_IO; }
        const uint16_t c = rd<uint16_t>(&b[off]);
        off += 2 + (size_t)c * 4;
        edges += c;
    }
    o->graph_edges = edges;
    *vec_off = align_to(off, ALIGN);
    if (*vec_off + (size_t)o->num_vectors * o->dimension * 4 > b.size()) { set_error("VAMA: truncated vector section"); return SHODH_ERR_IO; }
    return SHODH_OK;
}

// ---- SPAN ----
struct SpanOff { uint64_t centroids, codebook, pindex, pdata; };
int span_parse(const std::vector<unsigned char> &b, shodh_span_info *o, SpanOff *so) {
    if (b.size() < SPAN_HEADER) { set_error("SPAN: header too small"); return SHODH_ERR_IO; }
    if (memcmp(b.data(), "SPAN", 4) != 0) { set_error("SPAN: invalid magic bytes"); return SHODH_ERR_IO; }
    if (rd<uint32_t>(&b[4]) != 1) { set_error("SPAN: unsupported version %u", rd<uint32_t>(&b[4])); return SHODH_ERR_UNSUPPORTED; }
    o->num_vectors = rd<uint64_t>(&b[8]);
    o->num_partitions = rd<uint32_t>(&b[16]);
    o->dimension = rd<uint32_t>(&b[20]);
    o->pq_enabled = b[24];
    o->pq_subvectors = rd<uint32_t>(&b[25]);
    o->distance_metric = b[29];
    const uint64_t stored = rd<uint64_t>(&b[30]);
    so->centroids = rd<uint64_t>(&b[38]); so->codebook = rd<uint64_t>(&b[46]); so->pindex = rd<uint64_t>(&b[54]); so->pdata = rd<uint64_t>(&b[62]);
    if (stored != fnv1a64(b.data() + SPAN_HEADER, b.size() - SPAN_HEADER)) { set_error("SPAN: checksum mismatch"); return SHODH_ERR_IO; }
    const size_t P = o->num_partitions, D = o->dimension;
    if (so->centroids + P * D * 4 > b.size() || so->pindex + P * 12 > b.size()) { set_error("SPAN: truncated file"); return SHODH_ERR_IO; }
    o->pq_num_centroids = 0; o->pq_subvec_dim = 0;
    if (o->pq_enabled == 1) {
        if (so->codebook + 12 > b.size()) { set_error("SPAN: truncated codebook"); return SHODH_ERR_IO; }
        o->pq_subvectors = rd<uint32_t>(&b[so->codebook]);        // the section's own copy is what load_from_file trusts (spann.rs:933)
        o->pq_num_centroids = rd<uint32_t>(&b[so->codebook + 4]);
        o->pq_subvec_dim = rd<uint32_t>(&b[so->codebook + 8]);
        if (so->codebook + 12 + (size_t)o->pq_subvectors * o->pq_num_centroids * o->pq_subvec_dim * 4 > b.size()) { set_error("SPAN: truncated codebook"); return SHODH_ERR_IO; }
    }
    const size_t esz = 4 + (o->pq_enabled == 1 ? o->pq_subvectors : 0);
    uint64_t total = 0;
    for (size_t p = 0; p < P; ++p) {
        const uint64_t off = rd<uint64_t>(&b[so->pindex + p * 12]);
        const uint32_t cnt = rd<uint32_t>(&b[so->pindex + p * 12 + 8]);
        if (so->pdata + off + (uint64_t)cnt * esz > b.size()) { set_error("SPAN: posting list %zu out of bounds", p); return SHODH_ERR_IO; }
        total += cnt;
    }
    o->total_postings = total;
    return SHODH_OK;
}

}  // namespace
}  // namespace shodh

using namespace shodh;

extern "C" {

int shodh_vama_info_read(const char *path, shodh_vama_info *out) {
    if (!path || !out) { set_error("null argument"); return SHODH_ERR_INVALID; }
    std::vector<unsigned char> b;
    SHODH_TRY(read_file(path, b));
    size_t g, v;
    return vama_parse(b, out, &g, &v);
}

int shodh_vama_load(const char *path, float *vectors, uint32_t *deleted, uint16_t *degree, uint32_t *neighbors) {
    if (!path) { set_error("null argument"); return SHODH_ERR_INVALID; }
    std::vector<unsigned char> b;
    SHODH_TRY(read_file(path, b));
    shodh_vama_info o;
    size_t g, v;
    SHODH_TRY(vama_parse(b, &o, &g, &v));
    if (deleted) memcpy(deleted, &b[VAMA_HEADER], (size_t)o.deleted_count * 4);
    if (degree || neighbors) {
        size_t off = g, e = 0;
        for (uint64_t i = 0; i < o.num_vectors; ++i) {
            const uint16_t c = rd<uint16_t>(&b[off]);
            off += 2;
            if (degree) degree[i] = c;
            if (neighbors) memcpy(neighbors + e, &b[off], (size_t)c * 4);
            off += (size_t)c * 4;
            e += c;
        }
    }
    if (vectors) memcpy(vectors, &b[v], (size_t)o.num_vectors * o.dimension * 4);
    return SHODH_OK;
}

int shodh_vama_save(const char *path, const float *vectors, uint64_t n, uint32_t dim, uint32_t max_degree, uint32_t medoid,
                    uint8_t metric, const uint32_t *deleted, uint32_t deleted_count, uint64_t incremental_inserts,
                    const uint16_t *degree, const uint32_t *neighbors) {
    if (!path || (n && !vectors) || (deleted_count && !deleted) || (degree && !neighbors)) { set_error("null argument"); return SHODH_ERR_INVALID; }
    size_t graph = 0;
    for (uint64_t i = 0; i < n; ++i) graph += 2 + (size_t)(degree ? degree[i] : 0) * 4;
    const size_t voff = align_to(VAMA_HEADER + (size_t)deleted_count * 4 + graph, ALIGN);
    std::vector<unsigned char> b(voff + (size_t)n * dim * 4, 0);
    memcpy(&b[0], "VAMA", 4);
    wr<uint32_t>(&b[4], 1); wr<uint64_t>(&b[8], n); wr<uint32_t>(&b[16], dim); wr<uint32_t>(&b[20], max_degree); wr<uint32_t>(&b[24], medoid);
    b[28] = metric; wr<uint32_t>(&b[29], deleted_count); wr<uint64_t>(&b[33], incremental_inserts);
    size_t off = VAMA_HEADER;
    if (deleted_count) memcpy(&b[off], deleted, (size_t)deleted_count * 4);
    off += (size_t)deleted_count * 4;
    size_t e = 0;
    for (uint64_t i = 0; i < n; ++i) {
        const uint16_t c = degree ? degree[i] : 0;
        wr<uint16_t>(&b[off], c);
        off += 2;
        if (c) memcpy(&b[off], neighbors + e, (size_t)c * 4);
        off += (size_t)c * 4;
        e += c;
    }
    if (n) memcpy(&b[voff], vectors, (size_t)n * dim * 4);
    wr<uint64_t>(&b[41], fnv1a64(b.data() + VAMA_HEADER, b.size() - VAMA_HEADER));
    return write_file(path, b);
}

int shodh_span_info_read(const char *path, shodh_span_info *out) {
    if (!path || !out) { set_error("null argument"); return SHODH_ERR_INVALID; }
    std::vector<unsigned char> b;
    SHODH_TRY(read_file(path, b));
    SpanOff so;
    return span_parse(b, out, &so);
}

int shodh_span_load(const char *path, float *centroids, float *codebook, uint64_t *list_off, uint32_t *ids, uint8_t *codes) {
    if (!path) { set_error("null argument"); return SHODH_ERR_INVALID; }
    std::vector<unsigned char> b;
    SHODH_TRY(read_file(path, b));
    shodh_span_info o;
    SpanOff so;
    SHODH_TRY(span_parse(b, &o, &so));
    const size_t P = o.num_partitions, D = o.dimension;
    const bool pq = o.pq_enabled == 1;
    const size_t M = pq ? o.pq_subvectors : 0, esz = 4 + M;
    if (centroids) memcpy(centroids, &b[so.centroids], P * D * 4);
    if (codebook && pq) memcpy(codebook, &b[so.codebook + 12], (size_t)o.pq_subvectors * o.pq_num_centroids * o.pq_subvec_dim * 4);
    uint64_t at = 0;
    for (size_t p = 0; p < P; ++p) {
        const uint64_t off = rd<uint64_t>(&b[so.pindex + p * 12]);
        const uint32_t cnt = rd<uint32_t>(&b[so.pindex + p * 12 + 8]);
        if (list_off) list_off[p] = at;
        const unsigned char *src = &b[so.pdata + off];
        for (uint32_t i = 0; i < cnt; ++i) {
            if (ids) ids[at + i] = rd<uint32_t>(src + (size_t)i * esz);
            if (codes && M) memcpy(codes + (at + i) * M, src + (size_t)i * esz + 4, M);
        }
        at += cnt;
    }
    if (list_off) list_off[P] = at;
    return SHODH_OK;
}

int shodh_span_save(const char *path, uint64_t num_vectors, uint32_t P, uint32_t dim, uint32_t M, uint8_t metric, const float *centroids,
                    const float *codebook, const uint64_t *list_off, const uint32_t *ids, const uint8_t *codes) {
    if (!path || !centroids || !list_off || (list_off[P] && !ids)) { set_error("null argument"); return SHODH_ERR_INVALID; }
    if (P == 0) { set_error("Cannot save empty index"); return SHODH_ERR_STATE; }            // spann.rs:757-759
    const bool pq = codebook != nullptr;
    if (pq && (M != dim / 8 || !codes)) { set_error("SPAN: pq_subvectors must be dimension / 8 (spann.rs:765)"); return SHODH_ERR_INVALID; }
    const size_t coff = align_to(SPAN_HEADER, ALIGN), csz = (size_t)P * dim * 4;
    const size_t boff = align_to(coff + csz, ALIGN), bsz = pq ? 12 + (size_t)M * 256 * 8 * 4 : 0;
    const size_t ioff = align_to(boff + bsz, ALIGN), isz = (size_t)P * 12;
    const size_t doff = align_to(ioff + isz, ALIGN), esz = 4 + (pq ? M : 0);
    std::vector<unsigned char> b(doff + (size_t)list_off[P] * esz, 0);
    memcpy(&b[0], "SPAN", 4);
    wr<uint32_t>(&b[4], 1); wr<uint64_t>(&b[8], num_vectors); wr<uint32_t>(&b[16], P); wr<uint32_t>(&b[20], dim);
    b[24] = pq ? 1 : 0; wr<uint32_t>(&b[25], pq ? M : 0); b[29] = metric;
    wr<uint64_t>(&b[38], coff); wr<uint64_t>(&b[46], boff); wr<uint64_t>(&b[54], ioff); wr<uint64_t>(&b[62], doff);
    memcpy(&b[coff], centroids, csz);
    if (pq) {
        wr<uint32_t>(&b[boff], M); wr<uint32_t>(&b[boff + 4], 256); wr<uint32_t>(&b[boff + 8], 8);
        memcpy(&b[boff + 12], codebook, (size_t)M * 256 * 8 * 4);
    }
    for (uint32_t p = 0; p < P; ++p) {
        const uint64_t lo = list_off[p], cnt = list_off[p + 1] - lo;
        wr<uint64_t>(&b[ioff + (size_t)p * 12], lo * esz);
        wr<uint32_t>(&b[ioff + (size_t)p * 12 + 8], (uint32_t)cnt);
        unsigned char *dst = &b[doff + lo * esz];
        for (uint64_t i = 0; i < cnt; ++i) {
            wr<uint32_t>(dst + i * esz, ids[lo + i]);
            if (pq) memcpy(dst + i * esz + 4, codes + (lo + i) * M, M);
        }
    }
    wr<uint64_t>(&b[30], fnv1a64(b.data() + SPAN_HEADER, b.size() - SPAN_HEADER));
    return write_file(path, b);
}

}  // extern "C"
