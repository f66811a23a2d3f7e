/*
 * oracle/esdf_oracle.h -- CPU restatement of voxblox::EsdfIntegrator::
 * updateFromTsdfLayerBatch, which voxgraph reaches through
 * cblox::TsdfEsdfSubmap::generateEsdf() at
 * voxgraph/src/frontend/submap_collection/voxgraph_submap.cpp:86.
 * TEST INFRASTRUCTURE ONLY (see reg_oracle.h for the usage rule).
 *
 * PARITY UNPINNED: voxblox is not vendored in /root/reference and not pinned;
 * everything here is [recalled] from its public sources (integrator/
 * esdf_integrator.cc, utils/bucket_queue.h, utils/neighbor_tools.h): fixed band
 * |tsdf| < min_distance_m copied from the TSDF, the rest seeded at
 * sign * default_distance_m and lowered by a label-correcting wavefront over the
 * 26-neighbourhood (quasi-Euclidean step lengths 1, sqrt2, sqrt3 voxels), bucketed
 * priority queue (20 buckets, FIFO inside a bucket), updates only when they
 * improve by more than min_diff_m.
 */
#ifndef VOXGRAPH_AMD_ORACLE_ESDF_ORACLE_H_
#define VOXGRAPH_AMD_ORACLE_ESDF_ORACLE_H_
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

/* voxblox::EsdfIntegrator::Config defaults [recalled] */
typedef struct orc_esdf_config {
  float max_distance_m;     /* 2.0   */
  float min_distance_m;     /* 0.2   */
  float default_distance_m; /* 2.0   */
  float min_diff_m;         /* 0.001 */
  float min_weight;         /* 1e-6  */
  int num_buckets;          /* 20    */
} orc_esdf_config;
void orc_esdf_config_default(orc_esdf_config* c);

/* Layers in voxblox layout: block_index[n][3]; tsdf_* and esdf_* [n][vps^3].
 * Returns the number of wavefront updates, or -1 on allocation failure. */
int64_t orc_esdf_from_tsdf_batch(const orc_esdf_config* cfg, float voxel_size, int vps,
                                 int n_blocks, const int32_t* block_index,
                                 const float* tsdf_distance, const float* tsdf_weight,
                                 float* esdf_distance, uint8_t* esdf_observed);
#ifdef __cplusplus
}
#endif
#endif
