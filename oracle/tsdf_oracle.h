/*
 * oracle/tsdf_oracle.h -- CPU restatement of voxblox::FastTsdfIntegrator::
 * integratePointCloud (the TSDF path).  TEST INFRASTRUCTURE ONLY: only tests/,
 * __graft_entry__.smoke() and bench.py's cpu_baseline leg may link or call this.
 *
 * PARITY UNPINNED.  The arithmetic lives in voxblox, which is not vendored in
 * /root/reference (README.md:30 merely asks for it to be installed; no version
 * is pinned).  The reference holds only the call site and parameters:
 *   voxgraph/src/frontend/measurement_processors/pointcloud_integrator.cpp:66-83
 *   voxgraph/config/voxgraph_mapper.yaml:21-28
 * Everything below restates voxblox's published algorithm [recalled]
 * (integrator/tsdf_integrator.cc FastTsdfIntegrator, integrator/
 * integrator_utils.h RayCaster, utils/approx_hash_array.h) with
 * integrator_threads = 1, i.e. one of the orders the multi-threaded reference
 * may produce, in either integration_order_mode ("mixed", "sorted":
 * voxgraph/config/voxgraph_mapper.yaml:29).  There are no golden vectors for it anywhere.
 */
#ifndef VOXGRAPH_AMD_ORACLE_TSDF_ORACLE_H_
#define VOXGRAPH_AMD_ORACLE_TSDF_ORACLE_H_

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* voxblox::TsdfIntegratorBase::Config (defaults in comments; voxgraph_mapper.yaml
 * overrides in brackets) */
typedef struct orc_tsdf_config {
  float default_truncation_distance;      /* 0.1   [0.60] */
  float max_weight;                       /* 10000        */
  int voxel_carving_enabled;              /* 1            */
  float min_ray_length_m;                 /* 0.1          */
  float max_ray_length_m;                 /* 5.0   [16.0] */
  int use_const_weight;                   /* 0     [1]    */
  int allow_clear;                        /* 1            */
  int use_weight_dropoff;                 /* 1     [1]    */
  int use_sparsity_compensation_factor;   /* 0     [1]    */
  float sparsity_compensation_factor;     /* 1.0   [20.0] */
  float start_voxel_subsampling_factor;   /* 2.0          */
  int max_consecutive_ray_collisions;     /* 2            */
  int clear_checks_every_n_frames;        /* 1            */
  int integration_order;                  /* ThreadSafeIndexFactory [recalled]: 1 = "mixed" (default: 1024-point
                                           * groups round-robin), 2 = "sorted" (ascending f32 squaredNorm of
                                           * point_C; voxblox uses the unstable std::sort, the order at EQUAL
                                           * range is unspecified there: here ascending point index),
                                           * 0 = plain input order (not a voxblox mode; used by tests) */
  int enable_anti_grazing;                /* 0 (MergedTsdfIntegrator only)        */
} orc_tsdf_config;

void orc_tsdf_config_default(orc_tsdf_config* cfg);

typedef struct orc_tsdf_layer orc_tsdf_layer;           /* voxblox::Layer<TsdfVoxel> */
typedef struct orc_tsdf_integrator orc_tsdf_integrator; /* voxblox::FastTsdfIntegrator */

orc_tsdf_layer* orc_tsdf_layer_create(float voxel_size, int vps);
void orc_tsdf_layer_destroy(orc_tsdf_layer* L);
int orc_tsdf_layer_num_blocks(const orc_tsdf_layer* L);
/* block_index[n][3], distance/weight [n][vps^3], rgba [n][vps^3][4]; any may be NULL */
void orc_tsdf_layer_download(const orc_tsdf_layer* L, int32_t* block_index,
                             float* distance, float* weight, uint8_t* rgba);

orc_tsdf_integrator* orc_tsdf_integrator_create(const orc_tsdf_config* cfg,
                                                orc_tsdf_layer* layer);
void orc_tsdf_integrator_destroy(orc_tsdf_integrator* I);
void orc_tsdf_integrator_set_layer(orc_tsdf_integrator* I, orc_tsdf_layer* layer);

/* integratePointCloud(T_G_C, points_C, colors, freespace_points).
 * T_G_C = {qw,qx,qy,qz, tx,ty,tz} (voxblox::Transformation, f32);
 * points_C [n][3] in the sensor frame; rgba [n][4] (NULL = all zero).
 * Returns the number of voxel updates performed (updateTsdfVoxel calls). */
int64_t orc_tsdf_integrate(orc_tsdf_integrator* I, const float T_G_C[7],
                           const float* points_C, const uint8_t* rgba, int64_t n,
                           int freespace_points);

/* bench.py's cpu_baseline leg: `repeats` passes over `n_scans` scans (poses [n_scans][7], clouds of n points
 * each, concatenated) in one call, so that a timing loop over many host threads never goes back to the
 * interpreter between scans.  Returns the total number of voxel updates. */
int64_t orc_tsdf_integrate_sequence(orc_tsdf_integrator* I, int n_scans, const float* poses,
                                    const float* points_C, int64_t n, int repeats);

/* voxblox::MergedTsdfIntegrator::integratePointCloud [recalled, integrator/tsdf_integrator.cc]:
 * bundleRays groups the valid points by the voxel their end point falls in (clearing rays in a map of
 * their own), integrateVoxel merges a group into one weighted-mean point (running mean in visiting
 * order, weight = sum of the point weights; a clearing group keeps only its first point), casts ONE
 * ray for it through all its voxels (no early-out) and updates every voxel with the merged weight;
 * with enable_anti_grazing a ray skips voxels that are the end voxel of another group.  All surface
 * groups first, then all clearing groups.  Uses the same integrator object (config, layer); the
 * approximate sets are not involved.  Returns the number of voxel updates. */
int64_t orc_tsdf_merged_integrate(orc_tsdf_integrator* I, const float T_G_C[7],
                                  const float* points_C, const uint8_t* rgba, int64_t n,
                                  int freespace_points);

/* The single thread's own event log, in the racing kernel's format (include/voxgraph_amd_bench.h "event log"; every
 * update its own one-record fold): scans from now on append to buffer[0 .. capacity_words) (NULL: off).  For
 * tsdf_replay.h's checker: a sequential run is one legal interleaving. */
void orc_tsdf_integrator_set_log(orc_tsdf_integrator* I, uint64_t* buffer, int64_t capacity_words);
int64_t orc_tsdf_integrator_log_words(const orc_tsdf_integrator* I, int64_t* lost);
/* the two approximate sets as they are (2^20 words each, either may be NULL) and their offsets */
void orc_tsdf_integrator_download_sets(const orc_tsdf_integrator* I, uint64_t* start_set, uint64_t* observed_set,
                                       uint64_t offsets[2]);

#ifdef __cplusplus
}
#endif
#endif
