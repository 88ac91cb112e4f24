// temporary stubs (replaced by ivfpq.hip / encoder.hip)
#include "common.h"
namespace shodh {
struct IvfpqState {};
void ivfpq_destroy(IvfpqState *s) { delete s; }
int ivfpq_search(IvfpqState *, const shodh_index_cfg &, const float *, uint32_t, uint32_t, uint32_t *, float *, uint32_t *, hipStream_t) {
    set_error("IVF-PQ not built yet"); return SHODH_ERR_UNSUPPORTED; }
}
using namespace shodh;
extern "C" {
int shodh_index_set_ivfpq(shodh_index *, const float *, uint32_t, const float *, uint32_t, uint32_t, const uint64_t *, const uint32_t *, const uint8_t *) { set_error("unsupported"); return SHODH_ERR_UNSUPPORTED; }
int shodh_index_ivfpq_insert(shodh_index *, uint32_t, const float *) { set_error("unsupported"); return SHODH_ERR_UNSUPPORTED; }
int shodh_index_ivfpq_encode(shodh_index *, const float *, uint64_t, uint32_t *, uint8_t *) { set_error("unsupported"); return SHODH_ERR_UNSUPPORTED; }
int shodh_ivfpq_train(int, const float *, uint64_t, uint32_t, uint32_t, uint32_t, uint32_t, const uint32_t *, const uint32_t *, float *, float *) { set_error("unsupported"); return SHODH_ERR_UNSUPPORTED; }
int shodh_cosine_similarity_batch(int, const float *, const float *, uint64_t, uint32_t, uint32_t, float *) { set_error("unsupported"); return SHODH_ERR_UNSUPPORTED; }
void shodh_embed_cfg_default(shodh_embed_cfg *) {}
int shodh_embedder_create(const shodh_embed_cfg *, shodh_embedder **) { set_error("unsupported"); return SHODH_ERR_UNSUPPORTED; }
void shodh_embedder_destroy(shodh_embedder *) {}
uint64_t shodh_embedder_param_count(const shodh_embedder *) { return 0; }
int shodh_embedder_load_weights(shodh_embedder *, const float *, uint64_t) { return SHODH_ERR_UNSUPPORTED; }
int shodh_embedder_init_synthetic(shodh_embedder *, uint64_t, float *, uint64_t) { return SHODH_ERR_UNSUPPORTED; }
uint32_t shodh_embedder_dimension(const shodh_embedder *) { return 0; }
int shodh_embedder_encode_ids(shodh_embedder *, const int32_t *, const uint8_t *, uint32_t, float *) { return SHODH_ERR_UNSUPPORTED; }
int shodh_embedder_encode_ids_device(shodh_embedder *, const int32_t *, const uint8_t *, uint32_t, float *, void *) { return SHODH_ERR_UNSUPPORTED; }
int shodh_embedder_stage_timings(const shodh_embedder *, float *) { return SHODH_ERR_UNSUPPORTED; }
}
