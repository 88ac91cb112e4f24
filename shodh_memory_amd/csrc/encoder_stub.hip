// placeholder until encoder.hip lands (same ABI)
#include "common.h"
using namespace shodh;
extern "C" {
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
