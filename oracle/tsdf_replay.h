/*
 * oracle/tsdf_replay.h -- replay checker for the racing TSDF kernel's event log.  TEST INFRASTRUCTURE ONLY (only
 * tests/ may link or call this); parity unpinned like the oracle it is made of (tsdf_oracle.h).
 *
 * voxblox::FastTsdfIntegrator::integratePointCloud run by worker threads (the call of
 * voxgraph/src/frontend/measurement_processors/pointcloud_integrator.cpp:83) has no single result: rays race on two
 * approximate hash sets and on per-voxel locks.  What it DOES guarantee is that the result is an interleaving of the
 * sequential steps -- every set exchange atomic, every ray deciding from the values its own exchanges returned, every
 * voxel update applied to the value the previous one left.  The racing GPU kernel (voxgraph_amd/csrc/
 * vgx_tsdf_coop_kernel.h) can log every such step (include/voxgraph_amd_bench.h "event log"); this checker takes the log
 * of ONE scan, the scan's inputs and the state before and after it, and verifies with the oracle's own functions
 * (tsdf_oracle_impl.h) that the log is a legal interleaving, bit for bit:
 *
 *   A. start set -- every valid point exchanged the oracle's value (or skipped the exchange next to a lane holding the
 *      same value); per slot the (returned -> written) pairs chain into ONE path from the slot's content before the scan
 *      to its content after it; a ray is cast iff its exchange returned another value.
 *   B. rays -- every cast point set up the oracle's ray; its observed-set exchanges are the oracle's voxels in order,
 *      steps 0, 1, 2, ... without a gap; it stopped exactly where the oracle's rule stops it GIVEN THE VALUES IT GOT
 *      (more than max_consecutive_ray_collisions "already there" in a row), or walked to the ray's end.  Exchanges behind
 *      that stop are the kernel's stated liberty (a peeked slot changed before its exchange): counted and bounded, never
 *      hidden.  Per slot of the observed set: one path, as for the start set.
 *   C. voxels -- every fold applied updateTsdfVoxel for its records, in the listed order, to the word it says it folded
 *      over, and published exactly the word the oracle computes; per voxel the (folded over -> published) pairs chain
 *      into ONE path from the voxel before the scan to the voxel after it; colours likewise (their own compare-and-swap);
 *      the union of all folds' records is exactly the set of updates the rays of B must emit -- none missing, none twice.
 *   D. everything without events is unchanged: set slots, voxels, colours.
 */
#ifndef VOXGRAPH_AMD_ORACLE_TSDF_REPLAY_H_
#define VOXGRAPH_AMD_ORACLE_TSDF_REPLAY_H_

#include <stdint.h>

#include "tsdf_oracle.h"

#ifdef __cplusplus
extern "C" {
#endif

typedef struct orc_replay_layer {   /* vgx_tsdf_layer_download's arrays */
  int32_t n_blocks;
  const int32_t* block_index;       /* [n][3] */
  const float* distance;            /* [n][vps^3] */
  const float* weight;
  const uint8_t* rgba;              /* [n][vps^3][4] */
} orc_replay_layer;

typedef struct orc_replay_report {
  int64_t points, valid_points, start_exchanges, start_skips, rays_cast, rays_bad;
  int64_t observed_exchanges, overrun_exchanges, rays_with_overrun, max_overrun;
  int64_t rays_stopped_early, rays_walked_to_end;
  int64_t required_updates, fold_events, folds_published, folds_left_alone, fold_records, longest_fold;
  int64_t colour_writes;
  int64_t start_slots_touched, observed_slots_touched, voxels_touched, voxels_with_several_links;
  int64_t new_blocks;
  int64_t errors;                   /* 0 = the log is a legal interleaving */
  char first_error[400];
} orc_replay_report;

/* Returns 0 when the log is legal, else the number of violations found (first one described in the report).
 * start_offset / observed_offset: the offsets the scan's values carry (vgx_tsdf_integrator_download_sets AFTER the scan).
 * *_pre / *_post: both sets (2^20 words each) and the layer before and after the scan.  trace: the scan's log. */
int64_t orc_tsdf_replay_check(const orc_tsdf_config* cfg, float voxel_size, int vps, const float T_G_C[7],
                              const float* points_C, const uint8_t* rgba, int64_t n, int freespace_points,
                              uint64_t start_offset, uint64_t observed_offset,
                              const uint64_t* start_pre, const uint64_t* start_post,
                              const uint64_t* observed_pre, const uint64_t* observed_post,
                              const orc_replay_layer* layer_pre, const orc_replay_layer* layer_post,
                              const uint64_t* trace, int64_t n_words, orc_replay_report* report);

#ifdef __cplusplus
}
#endif
#endif
