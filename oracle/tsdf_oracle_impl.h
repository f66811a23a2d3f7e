/*
 * oracle/tsdf_oracle_impl.h -- the per-point and per-voxel steps of the TSDF restatement (tsdf_oracle.c), shared with
 * the replay checker (tsdf_replay.c) so that the checker judges the racing kernel's event log with the oracle's OWN
 * functions, not with a third restatement.  TEST INFRASTRUCTURE, parity unpinned, everything [recalled] from voxblox
 * (see tsdf_oracle.h).  Build with -ffp-contract=off.
 */
#ifndef VOXGRAPH_AMD_ORACLE_TSDF_ORACLE_IMPL_H_
#define VOXGRAPH_AMD_ORACLE_TSDF_ORACLE_IMPL_H_

#include <math.h>
#include <stdint.h>

#include "tsdf_oracle.h"

static const float kCoordinateEpsilon = 1e-6f; /* voxblox::kCoordinateEpsilon */

/* getGridIndexFromPoint's static_cast<IndexElement>(std::floor(x)) [recalled].  The reference's cast is undefined
 * for NaN and beyond the integer range; here -- and in the product, csrc/vgx_tsdf_internal.h grid_index -- it is
 * DEFINED: NaN -> 0, saturating at the 32-bit limits (a point 2^31 voxels away is a driver's "no return" code, not
 * a measurement; NaN points pass isPointValid in the reference too).  Everything within +-2^31 voxels: the cast. */
static inline int64_t grid_index(float x) {
  x = floorf(x);
  if (!(x == x)) return 0;
  if (x >= 2147483648.0f) return 2147483647;
  if (x < -2147483648.0f) return -2147483647 - 1;
  return (int64_t)x;
}
static const float kFloatEpsilon = 1e-6f;      /* voxblox::kFloatEpsilon */
static const float kEpsilon = 1e-6f;           /* voxblox::kEpsilon */


/* LongIndexHash: static_cast<unsigned int>(x + y*17191 + z*17191^2) on int64 */
static inline uint64_t long_index_hash(const int64_t idx[3]) {
  int64_t v = idx[0] + idx[1] * 17191 + idx[2] * (int64_t)(17191 * 17191);
  return (uint64_t)(uint32_t)v;
}


/* Eigen _transformVector + translation (kindr::minimal transform) */
static inline void transform_point(const float T[7], const float v[3], float out[3]) {
  float w = T[0], x = T[1], y = T[2], z = T[3];
  float uv[3] = {y * v[2] - z * v[1], z * v[0] - x * v[2], x * v[1] - y * v[0]};
  uv[0] += uv[0];
  uv[1] += uv[1];
  uv[2] += uv[2];
  float c[3] = {y * uv[2] - z * uv[1], z * uv[0] - x * uv[2], x * uv[1] - y * uv[0]};
  out[0] = (v[0] + w * uv[0] + c[0]) + T[4];
  out[1] = (v[1] + w * uv[1] + c[1]) + T[5];
  out[2] = (v[2] + w * uv[2] + c[2]) + T[6];
}

static inline float norm3(const float v[3]) { return sqrtf(v[0] * v[0] + v[1] * v[1] + v[2] * v[2]); }

static inline int signum(float x) { return (x > 0.0f) - (x < 0.0f); }

/* updateTsdfVoxel + computeDistance + blendTwoColors */
static inline void update_voxel(const orc_tsdf_config* c, float vs, const float origin[3], const float point_G[3],
                         const int64_t gidx[3], const uint8_t color[4], float weight,
                         float* v_dist, float* v_weight, uint8_t* v_rgba) {
  float voxel_center[3], v_voxel_origin[3], v_point_origin[3];
  for (int a = 0; a < 3; ++a) {
    /* getCenterPointFromGridIndex: (idx + 0.5) * grid_size, double product -> f32 */
    voxel_center[a] = (float)(((double)(float)gidx[a] + 0.5) * (double)vs);
    v_voxel_origin[a] = voxel_center[a] - origin[a];
    v_point_origin[a] = point_G[a] - origin[a];
  }
  float dist_G = norm3(v_point_origin);
  float dot = v_voxel_origin[0] * v_point_origin[0] + v_voxel_origin[1] * v_point_origin[1] +
              v_voxel_origin[2] * v_point_origin[2];
  float dist_G_V = dot / dist_G;
  float sdf = dist_G - dist_G_V;

  float updated_weight = weight;
  const float dropoff_epsilon = vs;
  if (c->use_weight_dropoff && sdf < -dropoff_epsilon) {
    updated_weight = weight * (c->default_truncation_distance + sdf) /
                     (c->default_truncation_distance - dropoff_epsilon);
    updated_weight = fmaxf(updated_weight, 0.0f);
  }
  if (c->use_sparsity_compensation_factor) {
    if (fabsf(sdf) < c->default_truncation_distance)
      updated_weight *= c->sparsity_compensation_factor;
  }
  const float new_weight = *v_weight + updated_weight;
  if (new_weight < kFloatEpsilon) return;
  const float new_sdf = (sdf * updated_weight + *v_dist * *v_weight) / new_weight;
  if (fabsf(sdf) < c->default_truncation_distance) {
    float first_weight = *v_weight, second_weight = updated_weight;
    float total = first_weight + second_weight;
    first_weight /= total;
    second_weight /= total;
    for (int k = 0; k < 4; ++k)
      v_rgba[k] = (uint8_t)roundf((float)v_rgba[k] * first_weight + (float)color[k] * second_weight);
  }
  *v_dist = (new_sdf > 0.0f) ? fminf(c->default_truncation_distance, new_sdf)
                             : fmaxf(-c->default_truncation_distance, new_sdf);
  *v_weight = fminf(c->max_weight, new_weight);
}


/* isPointValid: 0 = the point is skipped; *is_clearing as the fast and merged integrators use it */
static inline int point_is_valid(const orc_tsdf_config* c, const float point_C[3], int freespace_points, int* is_clearing) {
  const float ray_distance = norm3(point_C);
  if (ray_distance < c->min_ray_length_m) {
    return 0;
  } else if (ray_distance > c->max_ray_length_m) {
    if (c->allow_clear || freespace_points) *is_clearing = 1; else return 0;
  } else {
    *is_clearing = freespace_points;
  }
  return 1;
}

/* the cell of the start-voxel dedup: a grid start_voxel_subsampling_factor times finer than the voxels */
static inline void start_cell(const orc_tsdf_config* c, float voxel_size_inv, const float point_G[3], int64_t gidx[3]) {
  const float sub_inv = c->start_voxel_subsampling_factor * voxel_size_inv;
  for (int a = 0; a < 3; ++a) gidx[a] = grid_index(point_G[a] * sub_inv + kCoordinateEpsilon);
}

/* getVoxelWeight */
static inline float point_weight(const orc_tsdf_config* c, const float point_C[3]) {
  if (c->use_const_weight) return 1.0f;
  const float dist_z = fabsf(point_C[2]);
  return dist_z > kEpsilon ? 1.0f / (dist_z * dist_z) : 0.0f;
}

/* RayCaster(origin, point_G, is_clearing, carving, max_ray, voxel_size_inv, trunc, cast_from_origin = false) as the fast
 * integrator builds it: the walk runs from the far end of the ray back towards the sensor */
typedef struct {
  int64_t curr[3], ray_length_in_steps;
  int step_sign[3];
  float t_to_next[3], t_step[3];
  int bad; /* a NaN ray end: the reference's RayCaster would walk garbage; both sides skip the ray */
} orc_ray;

static inline void fast_ray_setup(const orc_tsdf_config* c, float vsi, const float origin[3], const float point_G[3],
                                  int is_clearing, orc_ray* r) {
  float d[3] = {point_G[0] - origin[0], point_G[1] - origin[1], point_G[2] - origin[2]};
  float len = norm3(d);
  float unit_ray[3] = {d[0] / len, d[1] / len, d[2] / len};
  float ray_start[3], ray_end[3];
  const float trunc = c->default_truncation_distance;
  if (is_clearing) {
    float ray_length = fminf(fmaxf(len - trunc, 0.0f), c->max_ray_length_m);
    for (int a = 0; a < 3; ++a) {
      ray_end[a] = origin[a] + unit_ray[a] * ray_length;
      ray_start[a] = c->voxel_carving_enabled ? origin[a] : ray_end[a];
    }
  } else {
    for (int a = 0; a < 3; ++a) {
      ray_end[a] = point_G[a] + unit_ray[a] * trunc;
      ray_start[a] = c->voxel_carving_enabled ? origin[a] : (point_G[a] - unit_ray[a] * trunc);
    }
  }
  /* cast_from_origin == false: setupRayCaster(end_scaled, start_scaled) */
  float start_scaled[3], end_scaled[3];
  for (int a = 0; a < 3; ++a) {
    start_scaled[a] = ray_end[a] * vsi;
    end_scaled[a] = ray_start[a] * vsi;
  }
  r->ray_length_in_steps = 0;
  r->bad = 0;
  for (int a = 0; a < 3; ++a)
    if (isnan(start_scaled[a]) || isnan(end_scaled[a])) r->bad = 1;
  if (r->bad) return;
  for (int a = 0; a < 3; ++a) {
    r->curr[a] = grid_index(start_scaled[a] + kCoordinateEpsilon);
    int64_t end_index = grid_index(end_scaled[a] + kCoordinateEpsilon);
    int64_t diff = end_index - r->curr[a];
    r->ray_length_in_steps += diff < 0 ? -diff : diff;
    float ray_scaled = end_scaled[a] - start_scaled[a];
    r->step_sign[a] = signum(ray_scaled);
    float corrected_step = (float)(r->step_sign[a] > 0 ? r->step_sign[a] : 0);
    float start_scaled_shifted = start_scaled[a] - (float)r->curr[a];
    float distance_to_boundary = corrected_step - start_scaled_shifted;
    /* voxblox divides by ray_scaled unguarded; a component that is exactly 0
     * never advances here (t = +inf) instead of producing NaN */
    if (ray_scaled == 0.0f) {
      r->t_to_next[a] = INFINITY;
      r->t_step[a] = INFINITY;
    } else {
      r->t_to_next[a] = distance_to_boundary / ray_scaled;
      r->t_step[a] = (float)r->step_sign[a] / ray_scaled;
    }
  }
}

/* RayCaster::nextRayIndex: returns the current voxel in v and advances (minCoeff: first minimum) */
static inline void ray_next(orc_ray* r, int64_t v[3]) {
  v[0] = r->curr[0];
  v[1] = r->curr[1];
  v[2] = r->curr[2];
  int t_min_idx = 0;
  if (r->t_to_next[1] < r->t_to_next[t_min_idx]) t_min_idx = 1;
  if (r->t_to_next[2] < r->t_to_next[t_min_idx]) t_min_idx = 2;
  r->curr[t_min_idx] += r->step_sign[t_min_idx];
  r->t_to_next[t_min_idx] += r->t_step[t_min_idx];
}

#endif
