// TSDF path: voxblox::FastTsdfIntegrator::integratePointCloud on gfx950
// (call site voxgraph/src/frontend/measurement_processors/pointcloud_integrator.cpp:83;
// arithmetic restated from voxblox [recalled], see oracle/tsdf_oracle.h).
//
// One thread per ray.  The reference's worker threads become 10^4..10^5 concurrent
// rays with the same shared state and the same primitives:
//   * the two approximate hash sets are arrays of 64-bit words updated with
//     atomic exchange (ApproxHashSet::replaceHash),
//   * the per-voxel mutex + read-modify-write becomes a 64-bit compare-and-swap
//     on the packed {distance, weight} voxel, i.e. each ray's update is applied
//     atomically in SOME order -- exactly the reference's multi-thread semantics,
//   * blocks are allocated on demand from a pre-zeroed pool through a dense
//     block lookup table (no hashing, no locks held across iterations).  The layer is
//     UNBOUNDED like voxblox::Layer: before every scan the host makes sure the table's box
//     covers everything the scan can reach (origin +- max ray + truncation) and that the pool
//     has room for every block in that reach, growing either on the stream if not, so a
//     kernel never meets a block it cannot allocate.
// HBM-latency / atomic bound: 16 B per input point + 24 B per voxel update
// (SURVEY.md 8d); no MFMA.
#include <algorithm>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <string>
#include <utility>
#include <vector>

#include <rocprim/rocprim.hpp>  // device radix sort (MergedTsdfIntegrator's bundleRays)

#include "vgx_tsdf_internal.h"

#pragma clang fp contract(off)

namespace vgx {



// updateTsdfVoxel + computeDistance + Color::blendTwoColors [recalled]
__device__ __forceinline__ void update_voxel(const TsdfLayerDev& L, const vgx_tsdf_config& c,
                                             size_t at, float ox, float oy, float oz, float gx,
                                             float gy, float gz, int vx, int vy, int vz,
                                             uint32_t color, float weight) {
  const float vs = L.voxel_size;
  // getCenterPointFromGridIndex: (idx + 0.5) * voxel_size
  float cx = ((float)vx + 0.5f) * vs, cy = ((float)vy + 0.5f) * vs, cz = ((float)vz + 0.5f) * vs;
  float vvx = cx - ox, vvy = cy - oy, vvz = cz - oz;
  float vpx = gx - ox, vpy = gy - oy, vpz = gz - oz;
  float dist_G = norm3(vpx, vpy, vpz);
  float dot = (vvx * vpx + vvy * vpy) + vvz * vpz;
  float dist_G_V = dot / dist_G;
  float sdf = dist_G - dist_G_V;
  float updated_weight = weight;
  const float trunc = c.default_truncation_distance;
  if (c.use_weight_dropoff && sdf < -vs) {
    updated_weight = weight * (trunc + sdf) / (trunc - vs);
    updated_weight = fmaxf(updated_weight, 0.0f);
  }
  if (c.use_sparsity_compensation_factor && fabsf(sdf) < trunc)
    updated_weight *= c.sparsity_compensation_factor;

  unsigned long long* addr = &L.voxels[at];
  unsigned long long old = __hip_atomic_load(addr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  float old_w;
  while (true) {
    float d = __uint_as_float((unsigned)(old & 0xffffffffull));
    old_w = __uint_as_float((unsigned)(old >> 32));
    float new_weight = old_w + updated_weight;
    if (new_weight < 1e-6f) return;  // kFloatEpsilon
    float new_sdf = (sdf * updated_weight + d * old_w) / new_weight;
    float nd = (new_sdf > 0.0f) ? fminf(trunc, new_sdf) : fmaxf(-trunc, new_sdf);
    float nw = fminf(c.max_weight, new_weight);
    unsigned long long prev = atomicCAS(addr, old, pack_voxel(nd, nw));
    if (prev == old) break;
    old = prev;
  }
  if (fabsf(sdf) < trunc) {
    // blend with the weight this update saw
    uint32_t* caddr = &L.rgba[at];
    uint32_t oc = __hip_atomic_load(caddr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    float total = old_w + updated_weight;
    float fw = old_w / total, sw = updated_weight / total;
    while (true) {
      uint32_t nc = 0;
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        float a = (float)((oc >> (8 * k)) & 0xffu), b = (float)((color >> (8 * k)) & 0xffu);
        nc |= ((uint32_t)(uint8_t)roundf(a * fw + b * sw)) << (8 * k);
      }
      uint32_t prev = atomicCAS(caddr, oc, nc);
      if (prev == oc) break;
      oc = prev;
    }
  }
}

// ---------------------------------------------------------------------------
// Software-pipelined voxel updates.  updateTsdfVoxel is a chain of dependent memory round trips
// (load {d,w} -> CAS {d,w} -> load colour -> CAS colour), and the ray walk adds the approximate
// set's exchange in front of it: up to five L2 round trips per DDA step, and the scan's kernel time
// is the LONGEST ray's chain.  A ray never visits a voxel twice, so the update of voxel k-1 does
// not depend on the walk's step k: the walk keeps three updates in flight, one per stage,
//   stage 1: load {d,w} and colour      (voxel k-1)
//   stage 2: CAS {d,w}                  (voxel k-2, expected value from its stage 1)
//   stage 3: CAS colour                 (voxel k-3, blend weight from its stage 2)
// and issues them together with step k's exchange: one round trip per step.  Values loaded a step
// earlier may be stale when another ray got in between; the CAS then fails and is retried, exactly
// as in the unpipelined loop.  Per ray the updates are still applied in walk order.
struct PendingUpdate {
  size_t at = 0;
  float sdf = 0.0f, w = 0.0f, old_w = 0.0f;
  unsigned long long expected = 0ull;
  uint32_t expected_color = 0u;
  bool active = false, blend = false;
};

// geometry of updateTsdfVoxel + computeDistance (same operations as update_voxel)
__device__ __forceinline__ PendingUpdate make_update(const TsdfLayerDev& L, const vgx_tsdf_config& c,
                                                     size_t at, float ox, float oy, float oz, float gx,
                                                     float gy, float gz, int vx, int vy, int vz,
                                                     float weight) {
  PendingUpdate u;
  const float vs = L.voxel_size;
  float cx = ((float)vx + 0.5f) * vs, cy = ((float)vy + 0.5f) * vs, cz = ((float)vz + 0.5f) * vs;
  float vvx = cx - ox, vvy = cy - oy, vvz = cz - oz;
  float vpx = gx - ox, vpy = gy - oy, vpz = gz - oz;
  float dist_G = norm3(vpx, vpy, vpz);
  float dot = (vvx * vpx + vvy * vpy) + vvz * vpz;
  float dist_G_V = dot / dist_G;
  float sdf = dist_G - dist_G_V;
  float updated_weight = weight;
  const float trunc = c.default_truncation_distance;
  if (c.use_weight_dropoff && sdf < -vs) {
    updated_weight = weight * (trunc + sdf) / (trunc - vs);
    updated_weight = fmaxf(updated_weight, 0.0f);
  }
  if (c.use_sparsity_compensation_factor && fabsf(sdf) < trunc)
    updated_weight *= c.sparsity_compensation_factor;
  u.at = at;
  u.sdf = sdf;
  u.w = updated_weight;
  u.blend = fabsf(sdf) < trunc;
  u.active = true;
  return u;
}

__device__ __forceinline__ unsigned long long blended_voxel(const vgx_tsdf_config& c, const PendingUpdate& u,
                                                            unsigned long long old, bool* skip) {
  const float trunc = c.default_truncation_distance;
  float d = __uint_as_float((unsigned)(old & 0xffffffffull));
  float old_w = __uint_as_float((unsigned)(old >> 32));
  float new_weight = old_w + u.w;
  *skip = new_weight < 1e-6f;  // kFloatEpsilon: updateTsdfVoxel returns without touching the voxel
  float new_sdf = (u.sdf * u.w + d * old_w) / new_weight;
  float nd = (new_sdf > 0.0f) ? fminf(trunc, new_sdf) : fmaxf(-trunc, new_sdf);
  float nw = fminf(c.max_weight, new_weight);
  return pack_voxel(nd, nw);
}


// One pipeline beat: issues every stage's memory operation, then consumes the results and
// shifts.  `s1` enters with an address only; leaves through s2 and s3.
__device__ __forceinline__ void pipeline_beat(const TsdfLayerDev& L, const vgx_tsdf_config& c, uint32_t color,
                                              PendingUpdate& s1, PendingUpdate& s2, PendingUpdate& s3) {
  // ---- issue ----
  unsigned long long e1 = 0ull, prev2 = 0ull, want2 = 0ull;
  uint32_t ec1 = 0u, prevc3 = 0u, wantc3 = 0u;
  bool skip2 = false;
  if (s1.active) {
    e1 = __hip_atomic_load(&L.voxels[s1.at], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (s1.blend) ec1 = __hip_atomic_load(&L.rgba[s1.at], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
  if (s2.active) {
    want2 = blended_voxel(c, s2, s2.expected, &skip2);
    if (!skip2) prev2 = atomicCAS(&L.voxels[s2.at], s2.expected, want2);
  }
  const bool do3 = s3.active && s3.blend;
  if (do3) {
    wantc3 = blended_color(s3.expected_color, color, s3.old_w, s3.w);
    prevc3 = atomicCAS(&L.rgba[s3.at], s3.expected_color, wantc3);
  }
  // ---- consume ----
  if (do3) {
    uint32_t oc = s3.expected_color;
    while (prevc3 != oc) {  // another ray blended in between: retry on what it left
      oc = prevc3;
      prevc3 = atomicCAS(&L.rgba[s3.at], oc, blended_color(oc, color, s3.old_w, s3.w));
    }
  }
  if (s2.active) {
    unsigned long long old = s2.expected;
    while (!skip2 && prev2 != old) {
      old = prev2;
      want2 = blended_voxel(c, s2, old, &skip2);
      if (!skip2) prev2 = atomicCAS(&L.voxels[s2.at], old, want2);
    }
    s2.old_w = __uint_as_float((unsigned)(old >> 32));  // the weight this update saw
    if (skip2) s2.active = false;                        // no colour blend either
  }
  if (s1.active) {
    s1.expected = e1;
    s1.expected_color = ec1;
  }
  // ---- shift ----
  s3 = s2;
  s2 = s1;
  s1.active = false;
}

template <bool PIPELINED>
__global__ __launch_bounds__(256) void tsdf_integrate_kernel(TsdfLayerDev L, TsdfIntegratorDev I,
                                                            float qw, float qx, float qy, float qz,
                                                            float tx, float ty, float tz,
                                                            const float* __restrict__ points_C,
                                                            const uint32_t* __restrict__ rgba,
                                                            long long n, int freespace_points) {
  const vgx_tsdf_config& c = I.cfg;
  long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  unsigned long long my_updates = 0, my_dropped = 0, my_walk = 0, my_blends = 0;
  if (i < n) {
    float px = points_C[3 * i], py = points_C[3 * i + 1], pz = points_C[3 * i + 2];
    uint32_t color = rgba ? rgba[i] : 0u;
    // isPointValid
    bool valid = true, is_clearing = false;
    float ray_distance = norm3(px, py, pz);
    if (ray_distance < c.min_ray_length_m) {
      valid = false;
    } else if (ray_distance > c.max_ray_length_m) {
      if (c.allow_clear || freespace_points) is_clearing = true; else valid = false;
    } else {
      is_clearing = freespace_points != 0;
    }
    if (valid) {
      // T_G_C * point_C: Eigen _transformVector + translation
      float uvx = qy * pz - qz * py, uvy = qz * px - qx * pz, uvz = qx * py - qy * px;
      uvx += uvx; uvy += uvy; uvz += uvz;
      float ccx = qy * uvz - qz * uvy, ccy = qz * uvx - qx * uvz, ccz = qx * uvy - qy * uvx;
      float gx = (px + qw * uvx + ccx) + tx;
      float gy = (py + qw * uvy + ccy) + ty;
      float gz = (pz + qw * uvz + ccz) + tz;
      const float vsi = L.voxel_size_inv;
      const float sub_inv = c.start_voxel_subsampling_factor * vsi;
      int sx = grid_index(gx * sub_inv + 1e-6f), sy = grid_index(gy * sub_inv + 1e-6f),
          sz = grid_index(gz * sub_inv + 1e-6f);
      if (approx_replace(I.start_set, I.start_offset, sx, sy, sz)) {
        // RayCaster(origin, point_G, is_clearing, carving, max_ray, vsi, trunc, cast_from_origin=false)
        float dx = gx - tx, dy = gy - ty, dz = gz - tz;
        float len = norm3(dx, dy, dz);
        float ux = dx / len, uy = dy / len, uz = dz / len;
        const float trunc = c.default_truncation_distance;
        float sxx, syy, szz, exx, eyy, ezz;  // ray_start, ray_end
        if (is_clearing) {
          float ray_length = fminf(fmaxf(len - trunc, 0.0f), c.max_ray_length_m);
          exx = tx + ux * ray_length; eyy = ty + uy * ray_length; ezz = tz + uz * ray_length;
          sxx = c.voxel_carving_enabled ? tx : exx;
          syy = c.voxel_carving_enabled ? ty : eyy;
          szz = c.voxel_carving_enabled ? tz : ezz;
        } else {
          exx = gx + ux * trunc; eyy = gy + uy * trunc; ezz = gz + uz * trunc;
          sxx = c.voxel_carving_enabled ? tx : (gx - ux * trunc);
          syy = c.voxel_carving_enabled ? ty : (gy - uy * trunc);
          szz = c.voxel_carving_enabled ? tz : (gz - uz * trunc);
        }
        // setupRayCaster(end_scaled, start_scaled): walk from the surface to the sensor
        float ss[3] = {exx * vsi, eyy * vsi, ezz * vsi};
        float es[3] = {sxx * vsi, syy * vsi, szz * vsi};
        bool bad = false;
        int curr[3], sign[3];
        float t_next[3], t_step[3];
        long long steps = 0;
#pragma unroll
        for (int a = 0; a < 3; ++a) {
          bad |= (ss[a] != ss[a]) | (es[a] != es[a]);
          curr[a] = grid_index(ss[a] + 1e-6f);
          int end_index = grid_index(es[a] + 1e-6f);
          int diff = end_index - curr[a];
          steps += diff < 0 ? -diff : diff;
          float ray_scaled = es[a] - ss[a];
          sign[a] = signum(ray_scaled);
          float corrected = (float)(sign[a] > 0 ? sign[a] : 0);
          float shifted = ss[a] - (float)curr[a];
          float dist_b = corrected - shifted;
          if (ray_scaled == 0.0f) {
            t_next[a] = INFINITY;
            t_step[a] = INFINITY;
          } else {
            t_next[a] = dist_b / ray_scaled;
            t_step[a] = (float)sign[a] / ray_scaled;
          }
        }
        if (!bad) {
          // getVoxelWeight
          float weight = 1.0f;
          if (!c.use_const_weight) {
            float dist_z = fabsf(pz);
            weight = dist_z > 1e-6f ? 1.0f / (dist_z * dist_z) : 0.0f;
          }
          int collisions = 0;
          const int vps = L.vps, shift = L.vps_shift, mask = vps - 1;
          int last_b[3] = {INT32_MIN, INT32_MIN, INT32_MIN}, last_slot = -1;
          PendingUpdate s1, s2, s3;
          for (long long step = 0; step <= steps; ++step) {
            int vx = curr[0], vy = curr[1], vz = curr[2];
            int m = 0;
            if (t_next[1] < t_next[m]) m = 1;
            if (t_next[2] < t_next[m]) m = 2;
            // (kept branch-free on the register arrays)
            curr[0] += m == 0 ? sign[0] : 0; curr[1] += m == 1 ? sign[1] : 0; curr[2] += m == 2 ? sign[2] : 0;
            t_next[0] += m == 0 ? t_step[0] : 0.0f; t_next[1] += m == 1 ? t_step[1] : 0.0f;
            t_next[2] += m == 2 ? t_step[2] : 0.0f;
            // ApproxHashSet::replaceHash, issued first; its result is consumed after the pending
            // updates of the previous voxels have been issued as well
            unsigned int h = (unsigned int)vx + (unsigned int)vy * 17191u + (unsigned int)vz * 295530481u;
            unsigned long long v = (unsigned long long)h + I.observed_offset;
            unsigned long long seen = atomicExch(&I.observed_set[v & kSetMask], v);
            ++my_walk;  // dependent exchanges on this ray: the scan's critical path is the longest such chain
            if (PIPELINED) pipeline_beat(L, c, color, s1, s2, s3);
            if (seen == v) ++collisions; else collisions = 0;
            if (collisions > c.max_consecutive_ray_collisions) break;
            int bx = vx >> shift, by = vy >> shift, bz = vz >> shift;  // floor division (vps = 2^shift)
            if (bx != last_b[0] || by != last_b[1] || bz != last_b[2]) {
              last_slot = get_or_allocate_block(L, bx, by, bz);
              last_b[0] = bx; last_b[1] = by; last_b[2] = bz;
            }
            if (last_slot < 0) {
              ++my_dropped;
              continue;
            }
            size_t at = (size_t)last_slot * ((size_t)vps * vps * vps) +
                        (size_t)((vx & mask) + vps * ((vy & mask) + vps * (vz & mask)));
            if (PIPELINED) {
              s1 = make_update(L, c, at, tx, ty, tz, gx, gy, gz, vx, vy, vz, weight);
              my_blends += s1.blend ? 1u : 0u;
            } else
              update_voxel(L, c, at, tx, ty, tz, gx, gy, gz, vx, vy, vz, color, weight);
            ++my_updates;
          }
          if (PIPELINED) {  // drain
            pipeline_beat(L, c, color, s1, s2, s3);
            pipeline_beat(L, c, color, s1, s2, s3);
            pipeline_beat(L, c, color, s1, s2, s3);
          }
        }
      }
    }
  }
  // one atomic per wave for the statistics
  unsigned long long my_total = my_walk;
#pragma unroll
  for (int off = 32; off >= 1; off >>= 1) {
    my_updates += __shfl_xor(my_updates, off, 64);
    my_dropped += __shfl_xor(my_dropped, off, 64);
    my_blends += __shfl_xor(my_blends, off, 64);
    my_total += __shfl_xor(my_total, off, 64);
    const unsigned long long other = __shfl_xor(my_walk, off, 64);
    my_walk = other > my_walk ? other : my_walk;
  }
  if ((threadIdx.x & 63) == 0) {
    if (my_updates) atomicAdd(I.n_updates, my_updates);
    if (my_dropped) atomicAdd(L.dropped, my_dropped);
    if (my_walk) {  // vgx_tsdf_integrator_walk_stats (bench header)
      atomicMax(I.n_updates + 1, my_walk);
      atomicAdd(I.n_updates + 2, my_total);
      if (my_blends) atomicAdd(I.n_updates + 3, my_blends);
    }
  }
}

// ---------------------------------------------------------------------------
// voxblox::MergedTsdfIntegrator [recalled, integrator/tsdf_integrator.cc]
// ---------------------------------------------------------------------------
// bundleRays groups the valid points by the voxel their end point falls in, integrateVoxel merges a
// group into one weighted-mean point and casts ONE ray for it.  On the device: every point gets the
// key {clearing bit, end voxel} in the reference's visiting order (MixedThreadSafeIndex), a stable
// radix sort brings the groups together with that order intact inside each group, one thread per
// group merges its points sequentially (the running mean is order dependent in f32, so the order is
// the reference's) and walks the ray, updating voxels with the same 64-bit CAS as the fast
// integrator.  Surface groups first, clearing groups in a second launch (integrateRays twice).
constexpr unsigned long long kMergedInvalid = ~0ull;
constexpr long long kMergedBias = 1ll << 20;  // 21 bits per axis

__device__ __forceinline__ unsigned long long merged_key(int x, int y, int z, bool clearing) {
  return ((unsigned long long)clearing << 63) | (((unsigned long long)(x + kMergedBias) & 0x1fffffull) << 42) |
         (((unsigned long long)(y + kMergedBias) & 0x1fffffull) << 21) | ((unsigned long long)(z + kMergedBias) & 0x1fffffull);
}

__global__ __launch_bounds__(256) void merged_bundle_kernel(vgx_tsdf_config c, float vsi, float qw, float qx, float qy,
                                                           float qz, float tx, float ty, float tz,
                                                           const float* __restrict__ points_C, long long n,
                                                           const uint32_t* __restrict__ order,
                                                           int freespace_points, unsigned long long* __restrict__ keys,
                                                           unsigned int* __restrict__ idx,
                                                           unsigned int* __restrict__ counters,
                                                           uint32_t* __restrict__ g_count,
                                                           unsigned long long* __restrict__ scan_ctr) {
  const long long seq = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (seq >= n) return;
  // (this scan's group counters, per-group ray lengths and kCtr* counters start from zero: three memsets saved)
  g_count[seq] = 0u;
  if (seq == 0) {
    g_count[n] = 0u;
    counters[0] = counters[1] = counters[2] = counters[3] = 0u;
#pragma unroll
    for (int i = 0; i < kCtrCount; ++i) scan_ctr[i] = 0ull;
  }
  const long long pi = visiting_order_point(order, seq, n);  // ThreadSafeIndex: "mixed" or "sorted"
  const float px = points_C[3 * pi], py = points_C[3 * pi + 1], pz = points_C[3 * pi + 2];
  bool valid = true, is_clearing = false;
  const float ray_distance = norm3(px, py, pz);
  if (ray_distance < c.min_ray_length_m) {
    valid = false;
  } else if (ray_distance > c.max_ray_length_m) {
    if (c.allow_clear || freespace_points) is_clearing = true; else valid = false;
  } else {
    is_clearing = freespace_points != 0;
  }
  float uvx = qy * pz - qz * py, uvy = qz * px - qx * pz, uvz = qx * py - qy * px;
  uvx += uvx; uvy += uvy; uvz += uvz;
  const float ccx = qy * uvz - qz * uvy, ccy = qz * uvx - qx * uvz, ccz = qx * uvy - qy * uvx;
  const float gx = (px + qw * uvx + ccx) + tx, gy = (py + qw * uvy + ccy) + ty, gz = (pz + qw * uvz + ccz) + tz;
  const int vx = grid_index(gx * vsi + 1e-6f), vy = grid_index(gy * vsi + 1e-6f), vz = grid_index(gz * vsi + 1e-6f);
  // The key holds 21 bits per axis and WRAPS beyond, as the oracle's does (a point beyond +-2^20 voxels is a clearing
  // ray or nothing; its ray is clipped to max_ray_length_m and walks next to the sensor -- it must not be lost: a
  // freespace scan with a 276 km return lost 859 updates here until round 4, profiles/probes/merged_free_repro.py).
  // The one key that cannot be told from the invalid marker -- a clearing point whose three fields are all ones --
  // is reported (counters[5], folded into the scan's error word by merged_merge_kernel -- this kernel's thread 0 is
  // zeroing that word): the scan is refused.
  const unsigned long long key = merged_key(vx, vy, vz, is_clearing);
  if (valid && key == kMergedInvalid) counters[5] = 1u;
  keys[seq] = valid ? key : kMergedInvalid;
  idx[seq] = (unsigned int)pi;
}

// After the sort: group heads.  The heads are ranked by a prefix sum (group g = the g-th distinct key, so
// groups are numbered in key order: surface groups first, then clearing groups -- the order the
// reference's single thread walks its two voxel maps in, oracle/tsdf_oracle.c), and
//   counters[0] = number of groups            counters[1] = number of surface (non-clearing) entries
//   counters[2] = number of surface groups    counters[3] = number of valid entries
// (the ranks -- the exclusive prefix of "is a group's first entry" -- are taken inside the launch: TileChain; tiles of
// 1024 sorted entries, four consecutive ones per thread)
constexpr int kHeadsIpt = 4;
__global__ __launch_bounds__(256) void merged_heads_kernel(const unsigned long long* __restrict__ keys, long long n,
                                                          TileChain chain, unsigned int* __restrict__ group_start,
                                                          unsigned int* __restrict__ counters) {
  __shared__ uint32_t sh_word, sh4[4];
  const uint32_t tile = chain_tile(chain, &sh_word);
  const long long base = ((long long)tile * 256 + threadIdx.x) * kHeadsIpt;
  unsigned long long k[kHeadsIpt + 2];  // k[0]: the entry before this thread's, k[kHeadsIpt + 1]: the one after
#pragma unroll
  for (int e = 0; e < kHeadsIpt + 2; ++e) {
    const long long i = base - 1 + e;
    k[e] = (i >= 0 && i < n) ? keys[i] : kMergedInvalid;
  }
  uint32_t head[kHeadsIpt], mine = 0;
#pragma unroll
  for (int e = 0; e < kHeadsIpt; ++e) {
    const long long i = base + e;
    head[e] = (i < n && k[e + 1] != kMergedInvalid && (i == 0 || k[e] != k[e + 1])) ? 1u : 0u;
    mine += head[e];
  }
  uint32_t in_tile = 0;
  const uint32_t before = block_exclusive_sum(mine, sh4, in_tile);
  uint32_t rank = chain_exclusive_sum(chain, tile, in_tile, &sh_word) + before;
#pragma unroll
  for (int e = 0; e < kHeadsIpt; ++e) {
    const long long i = base + e;
    if (i >= n) break;
    const unsigned long long key = k[e + 1];
    const bool valid = key != kMergedInvalid;
    if (head[e]) group_start[rank] = (unsigned int)i;
    const bool next_valid = i + 1 < n && k[e + 2] != kMergedInvalid;
    if (valid && !next_valid) {  // the last valid entry
      counters[0] = rank + head[e];
      counters[3] = (unsigned int)(i + 1);
    }
    if (valid && !(key >> 63)) {
      const bool next_surface = next_valid && !(k[e + 2] >> 63);
      if (!next_surface) {  // the last surface entry
        counters[1] = (unsigned int)(i + 1);
        counters[2] = rank + head[e];
      }
    }
    rank += head[e];
  }
}

enum { kGroupValid = 1u, kGroupClearing = 2u };

// integrateVoxel's merge: L lanes per group load L of its points at a time; the running weighted mean
// is order dependent in f32, so the chain itself runs in the reference's visiting order (the stable
// sort kept it inside a group), fed by lane broadcasts.  Also the group's complete ray length.
template <int L>
__global__ __launch_bounds__(256) void merged_merge_kernel(vgx_tsdf_config c, float vsi, float qw, float qx, float qy,
                                                          float qz, float tx, float ty, float tz,
                                                          const float* __restrict__ points_C,
                                                          const uint32_t* __restrict__ rgba,
                                                          const unsigned long long* __restrict__ keys,
                                                          const unsigned int* __restrict__ idx,
                                                          const unsigned int* __restrict__ group_start,
                                                          const unsigned int* __restrict__ counters,
                                                          float4* __restrict__ g_pg, uint32_t* __restrict__ g_color,
                                                          uint32_t* __restrict__ g_flags,
                                                          uint32_t* __restrict__ g_count,
                                                          unsigned int* __restrict__ key_corner,
                                                          unsigned long long* __restrict__ scan_ctr) {
  const int lane = threadIdx.x & (L - 1);
  const unsigned int n_sub = gridDim.x * (blockDim.x / L);
  if (blockIdx.x == 0 && threadIdx.x == 0 && *key_corner) {  // merged_bundle_kernel: a key equal to the invalid marker
    scan_ctr[kCtrError] = kErrMergedKeyCorner;
    *key_corner = 0u;
  }
  unsigned long long my_steps = 0;  // ray steps of the groups this thread finished (lane 0 of a group)
  const unsigned int G = counters[0], n_valid = counters[3];
  if (blockIdx.x == 0 && threadIdx.x == 0) {  // (for the host: how many lanes a group deserves in the next scan)
    scan_ctr[kCtrGroups] = G;
    scan_ctr[kCtrGroupPoints] = n_valid;
  }
  for (unsigned int g = blockIdx.x * (blockDim.x / L) + threadIdx.x / L; g < G; g += n_sub) {
    const unsigned int i0 = group_start[g], i1 = g + 1 < G ? group_start[g + 1] : n_valid;
    const bool clearing_ray = (keys[i0] >> 63) != 0;
    float mx = 0.0f, my = 0.0f, mz = 0.0f, mw = 0.0f;
    uint32_t mcol = 0u;
    bool done = false;
    for (unsigned int b = i0; b < i1 && !done; b += L) {
      float px = 0.0f, py = 0.0f, pz = 0.0f, pw = 0.0f;
      uint32_t pc = 0u;
      if (b + lane < i1) {
        const unsigned int pi = idx[b + lane];
        px = points_C[3 * (size_t)pi]; py = points_C[3 * (size_t)pi + 1]; pz = points_C[3 * (size_t)pi + 2];
        pc = rgba ? rgba[pi] : 0u;
        pw = 1.0f;
        if (!c.use_const_weight) {  // getVoxelWeight
          const float dist_z = fabsf(pz);
          pw = dist_z > 1e-6f ? 1.0f / (dist_z * dist_z) : 0.0f;
        }
      }
      const int m = (int)min((unsigned int)L, i1 - b);
      for (int j = 0; j < m; ++j) {
        const float x = __shfl(px, j, L), y = __shfl(py, j, L), z = __shfl(pz, j, L), w = __shfl(pw, j, L);
        const uint32_t col = (uint32_t)__shfl((int)pc, j, L);
        if (w < 1e-6f) continue;  // kEpsilon
        const float total = mw + w;
        mx = (mx * mw + x * w) / total;
        my = (my * mw + y * w) / total;
        mz = (mz * mw + z * w) / total;
        mcol = blended_color(mcol, col, mw, w);
        mw += w;
        if (clearing_ray) {  // only the first point of a clearing group
          done = true;
          break;
        }
      }
    }
    if (lane == 0) {
      uint32_t flags = clearing_ray ? kGroupClearing : 0u, count = 0u;
      float gx = 0.0f, gy = 0.0f, gz = 0.0f;
      if (mw != 0.0f) {  // else every update would leave its voxel unchanged
        transform_point(qw, qx, qy, qz, tx, ty, tz, mx, my, mz, gx, gy, gz);
        // RayCaster(origin, merged_point_G, clearing_ray, carving, max_ray, vsi, trunc): cast_from_origin = true
        const RayDda r = ray_setup(c, vsi, tx, ty, tz, gx, gy, gz, clearing_ray, true);
        if (!r.bad && r.steps + 1 < (1ll << 24)) {
          flags |= kGroupValid;
          count = (uint32_t)(r.steps + 1);
        } else if (!r.bad) {
          scan_ctr[kCtrError] = kErrMergedRayTooLong;  // the step index is packed into 24 bits: the scan is refused (as the fast path does)
        }
      }
      g_pg[g] = make_float4(gx, gy, gz, mw);
      g_color[g] = mcol;
      g_flags[g] = flags;
      g_count[g] = count;
      my_steps += count;
    }
  }
  // the scan's ray steps, in 64 bits (the host sizes the write-out by it; the 32-bit offsets would wrap silently)
#pragma unroll
  for (int d = 32; d > 0; d >>= 1) my_steps += __shfl_down(my_steps, d);
  if ((threadIdx.x & 63) == 0 && my_steps) atomicAdd(&scan_ctr[kCtrTotal], my_steps);
}

__global__ __launch_bounds__(256) void tsdf_unpack_kernel(const unsigned long long* __restrict__ voxels,
                                                         size_t n, float* __restrict__ distance,
                                                         float* __restrict__ weight) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  unsigned long long v = voxels[i];
  distance[i] = __uint_as_float((unsigned)(v & 0xffffffffull));
  weight[i] = __uint_as_float((unsigned)(v >> 32));
}

// re-boxing: entries of the old block table move to their place in the new (larger) one;
// "pool exhausted" marks (-3) become free again, the pool has been enlarged meanwhile
__global__ void tsdf_lut_remap_kernel(const int32_t* __restrict__ old_lut, int3 old_min, int3 old_dim,
                                      int32_t* __restrict__ new_lut, int3 new_min, int3 new_dim) {
  const long long cells = (long long)old_dim.x * old_dim.y * old_dim.z;
  const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= cells) return;
  const int x = (int)(i % old_dim.x), y = (int)((i / old_dim.x) % old_dim.y), z = (int)(i / ((long long)old_dim.x * old_dim.y));
  const int slot = old_lut[i];
  if (slot < 0) return;
  const int nx = x + old_min.x - new_min.x, ny = y + old_min.y - new_min.y, nz = z + old_min.z - new_min.z;
  new_lut[nx + (long long)new_dim.x * (ny + (long long)new_dim.y * nz)] = slot;
}

__global__ void tsdf_lut_clear_exhausted_kernel(int32_t* __restrict__ lut, long long cells) {
  const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < cells && lut[i] == -3) lut[i] = -1;
}

__global__ void tsdf_lut_from_blocks_kernel(const int32_t* __restrict__ block_index, int n, int3 lut_min, int3 lut_dim,
                                            int32_t* __restrict__ lut) {
  const int b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= n) return;
  const int x = block_index[3 * b] - lut_min.x, y = block_index[3 * b + 1] - lut_min.y, z = block_index[3 * b + 2] - lut_min.z;
  lut[x + (long long)lut_dim.x * (y + (long long)lut_dim.y * z)] = b;
}

__global__ __launch_bounds__(256) void tsdf_pack_kernel(const float* __restrict__ distance,
                                                       const float* __restrict__ weight, size_t n,
                                                       unsigned long long* __restrict__ voxels) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  voxels[i] = pack_voxel(distance[i], weight[i]);
}

}  // namespace vgx

using namespace vgx;


extern "C" {

void vgx_tsdf_config_default(vgx_tsdf_config* c) {
  if (!c) return;
  c->default_truncation_distance = 0.1f;
  c->max_weight = 10000.0f;
  c->voxel_carving_enabled = 1;
  c->min_ray_length_m = 0.1f;
  c->max_ray_length_m = 5.0f;
  c->use_const_weight = 0;
  c->allow_clear = 1;
  c->use_weight_dropoff = 1;
  c->use_sparsity_compensation_factor = 0;
  c->sparsity_compensation_factor = 1.0f;
  c->start_voxel_subsampling_factor = 2.0f;
  c->max_consecutive_ray_collisions = 2;
  c->clear_checks_every_n_frames = 1;
  c->enable_anti_grazing = 0;
  c->deterministic = 0;
  c->integration_order = VGX_TSDF_ORDER_MIXED;
}

// ---------------------------------------------------------------------------
// layer storage: (re)allocation helpers, all on the context's stream
// ---------------------------------------------------------------------------
namespace {

constexpr int32_t kDefaultPoolBlocks = 256;

size_t voxels_per_block(const TsdfLayerDev& d) { return (size_t)d.vps * d.vps * d.vps; }

// New pool of `blocks` blocks: the first `used` blocks keep their contents, the rest is zero
// (a fresh voxblox block: distance 0, weight 0, colour 0).  Stream-ordered; the old arrays are
// released once the copies have run.
int grow_pool(vgx_tsdf_layer L, int32_t blocks, int32_t used) {
  vgx_ctx ctx = L->ctx;
  TsdfLayerDev& d = L->dev;
  const size_t vpb = voxels_per_block(d);
  unsigned long long* voxels = nullptr;
  uint32_t* rgba = nullptr;
  int32_t* block_index = nullptr;
  bool ok = hipMalloc(&voxels, (size_t)blocks * vpb * 8) == hipSuccess &&
            hipMalloc(&rgba, (size_t)blocks * vpb * 4) == hipSuccess &&
            hipMalloc(&block_index, (size_t)blocks * 12) == hipSuccess;
  if (ok) {
    const size_t keep = (size_t)used * vpb;
    ok = (keep == 0 || (hipMemcpyAsync(voxels, d.voxels, keep * 8, hipMemcpyDeviceToDevice, ctx->tsdf_stream) == hipSuccess &&
                        hipMemcpyAsync(rgba, d.rgba, keep * 4, hipMemcpyDeviceToDevice, ctx->tsdf_stream) == hipSuccess &&
                        hipMemcpyAsync(block_index, d.block_index, (size_t)used * 12, hipMemcpyDeviceToDevice, ctx->tsdf_stream) == hipSuccess)) &&
         hipMemsetAsync(voxels + keep, 0, ((size_t)blocks * vpb - keep) * 8, ctx->tsdf_stream) == hipSuccess &&
         hipMemsetAsync(rgba + keep, 0, ((size_t)blocks * vpb - keep) * 4, ctx->tsdf_stream) == hipSuccess &&
         hipStreamSynchronize(ctx->tsdf_stream) == hipSuccess;
  }
  if (!ok) {
    void* fresh[] = {voxels, rgba, block_index};
    for (void* q : fresh)
      if (q) (void)hipFree(q);
    (void)hipGetLastError();
    return set_error(ctx, VGX_ERR_NOMEM, "TSDF layer: block pool allocation failed (" + std::to_string(blocks) + " blocks)");
  }
  void* old[] = {d.voxels, d.rgba, d.block_index};
  for (void* q : old)
    if (q) (void)hipFree(q);
  d.voxels = voxels;
  d.rgba = rgba;
  d.block_index = block_index;
  d.max_blocks = blocks;
  return VGX_OK;
}

// New block table over [mn, mn + dm): existing entries move over, everything else is free.
int rebox(vgx_tsdf_layer L, const int32_t mn[3], const int32_t dm[3]) {
  vgx_ctx ctx = L->ctx;
  TsdfLayerDev& d = L->dev;
  size_t cells = 1;
  for (int a = 0; a < 3; ++a) cells *= (size_t)dm[a];
  if (cells > ((size_t)1 << 28))
    return set_error(ctx, VGX_ERR_UNSUPPORTED, "TSDF layer: block box exceeds 2^28 cells (dense lookup table)");
  int32_t* lut = nullptr;
  if (hipMalloc(&lut, cells * 4) != hipSuccess)
    return set_error(ctx, VGX_ERR_NOMEM, "TSDF layer: block table allocation failed");
  hipError_t e = hipMemsetAsync(lut, 0xff, cells * 4, ctx->tsdf_stream);
  if (e == hipSuccess && d.lut && L->lut_cells > 0) {
    hipLaunchKernelGGL(tsdf_lut_remap_kernel, dim3((unsigned)((L->lut_cells + 255) / 256)), dim3(256), 0, ctx->tsdf_stream,
                       d.lut, make_int3(d.lut_min[0], d.lut_min[1], d.lut_min[2]),
                       make_int3(d.lut_dim[0], d.lut_dim[1], d.lut_dim[2]), lut, make_int3(mn[0], mn[1], mn[2]),
                       make_int3(dm[0], dm[1], dm[2]));
    e = hipGetLastError();
  }
  if (e == hipSuccess) e = hipStreamSynchronize(ctx->tsdf_stream);
  if (e != hipSuccess) {
    (void)hipFree(lut);
    return set_error(ctx, VGX_ERR_HIP, std::string("TSDF layer: re-boxing failed: ") + hipGetErrorString(e));
  }
  if (d.lut) (void)hipFree(d.lut);
  d.lut = lut;
  L->lut_cells = cells;
  for (int a = 0; a < 3; ++a) {
    d.lut_min[a] = mn[a];
    d.lut_dim[a] = dm[a];
  }
  return VGX_OK;
}

// exact counters, waiting for the stream (rare: growth decisions, stats, download)
int read_stats_sync(vgx_tsdf_layer L, TsdfStats* out) {
  vgx_ctx ctx = L->ctx;
  VGX_HIP(ctx, hipSetDevice(ctx->device));
  VGX_HIP(ctx, hipStreamSynchronize(ctx->tsdf_stream));
  VGX_HIP(ctx, hipMemcpy(out, L->d_stats, sizeof(TsdfStats), hipMemcpyDeviceToHost));
  L->known_blocks = out->n_blocks;
  L->known_seq = L->scan_seq;
  L->recent.clear();
  L->readback_inflight = false;
  L->dropped_seen = out->dropped;
  return VGX_OK;
}

// non-blocking: adopt the counters an earlier asynchronous read-back has delivered meanwhile
void poll_readback(vgx_tsdf_layer L) {
  if (!L->readback_inflight || hipEventQuery(L->readback_done) != hipSuccess) return;
  L->readback_inflight = false;
  L->known_blocks = L->h_stats->n_blocks;
  L->known_seq = L->inflight_seq;
  L->dropped_seen = L->h_stats->dropped;
  size_t keep = 0;
  for (auto& r : L->recent)
    if (r.first > L->known_seq) L->recent[keep++] = r;
  L->recent.resize(keep);
}

// Before a scan whose rays start at `origin` (layer frame) and reach at most `reach` metres: the
// block table covers every block the scan can touch and the pool can hold all of them.
int reserve_for_scan(vgx_tsdf_layer L, const float origin[3], float reach) {
  vgx_ctx ctx = L->ctx;
  TsdfLayerDev& d = L->dev;
  poll_readback(L);
  if (L->dropped_seen != 0)
    return set_error(ctx, VGX_ERR_NOMEM, "TSDF layer: " + std::to_string(L->dropped_seen) +
                                             " voxel updates were dropped by an earlier scan (allocation failed)");
  const float bs = (float)d.vps * d.voxel_size, bs_inv = 1.0f / bs;
  int32_t lo[3], hi[3];
  for (int a = 0; a < 3; ++a) {
    lo[a] = (int32_t)std::floor((origin[a] - reach) * bs_inv) - 1;
    hi[a] = (int32_t)std::floor((origin[a] + reach) * bs_inv) + 1;
  }
  // upper bound on the blocks this scan can allocate: the blocks that come within `reach` of the
  // origin, WHEREVER in its block the origin sits (a block whose box comes within reach of some point of
  // the unit block around the origin: box-to-box distance).  A function of `reach` alone, so it is
  // computed once per reach and block size, not per scan (the per-scan triple loop grew as
  // (reach / block size)^3).
  if (L->bound_reach != reach) {
    const int32_t r = (int32_t)std::ceil(reach * bs_inv) + 1;
    int64_t count = 0;
    const float lim = (reach + bs * 0.01f) * (reach + bs * 0.01f);
    for (int32_t z = -r; z <= r; ++z)
      for (int32_t y = -r; y <= r; ++y)
        for (int32_t x = -r; x <= r; ++x) {
          const int32_t b[3] = {x, y, z};
          float d2 = 0.0f;
          for (int a = 0; a < 3; ++a) {
            // gap between block b[a] and the origin's own block (index 0) along this axis
            const float g = b[a] > 0 ? (float)(b[a] - 1) * bs : (b[a] < 0 ? (float)(-b[a] - 1) * bs : 0.0f);
            d2 += g * g;
          }
          if (d2 <= lim) ++count;
        }
    L->bound_reach = reach;
    L->bound_blocks = count;
  }
  const int64_t bound = L->bound_blocks;
  // 1. the box
  bool inside = L->lut_cells > 0;
  for (int a = 0; a < 3 && inside; ++a) inside = lo[a] >= d.lut_min[a] && hi[a] < d.lut_min[a] + d.lut_dim[a];
  if (!inside) {
    int32_t mn[3], dm[3];
    for (int a = 0; a < 3; ++a) {
      int32_t nlo = lo[a], nhi = hi[a];
      if (L->lut_cells > 0) {
        nlo = std::min(nlo, d.lut_min[a]);
        nhi = std::max(nhi, d.lut_min[a] + d.lut_dim[a] - 1);
      }
      // half a reach of slack on the side that grew: a moving sensor re-boxes every few metres, not every scan
      const int32_t slack = std::max<int32_t>(2, (hi[a] - lo[a]) / 4);
      if (L->lut_cells == 0 || nlo < d.lut_min[a]) nlo -= slack;
      if (L->lut_cells == 0 || nhi > d.lut_min[a] + d.lut_dim[a] - 1) nhi += slack;
      mn[a] = nlo;
      dm[a] = nhi - nlo + 1;
    }
    int rc = rebox(L, mn, dm);
    if (rc != VGX_OK) return rc;
    ++L->growths;
  }
  // 2. the pool: known + what earlier, not yet reported scans may have taken + this scan's reach
  int64_t pending = 0;
  for (auto& r : L->recent) pending += r.second;
  if (L->known_blocks + pending + bound > (int64_t)d.max_blocks) {
    TsdfStats st{};
    int rc = read_stats_sync(L, &st);  // exact count; also drains the stream
    if (rc != VGX_OK) return rc;
    if (st.dropped != 0)
      return set_error(ctx, VGX_ERR_NOMEM, "TSDF layer: voxel updates were dropped (allocation failed)");
    if ((int64_t)st.n_blocks + bound > (int64_t)d.max_blocks) {
      // room for this scan and fifteen more of the same reach queued behind it before the host has
      // to wait for the true count (at sensor rates it learns it from the asynchronous read-backs
      // long before), at least doubling.  48 KB per block: ~1 GB for 16 m LiDAR rays at 0.2 m.
      const int64_t want = std::max<int64_t>(2 * (int64_t)d.max_blocks, (int64_t)st.n_blocks + 16 * bound);
      if (want > INT32_MAX) return set_error(ctx, VGX_ERR_UNSUPPORTED, "TSDF layer: more than 2^31 blocks");
      rc = grow_pool(L, (int32_t)want, st.n_blocks);
      if (rc != VGX_OK) return rc;
      hipLaunchKernelGGL(tsdf_lut_clear_exhausted_kernel, dim3((unsigned)((L->lut_cells + 255) / 256)), dim3(256), 0,
                         ctx->tsdf_stream, d.lut, (long long)L->lut_cells);
      VGX_HIP(ctx, hipGetLastError());
      ++L->growths;
    }
  }
  L->recent.emplace_back(++L->scan_seq, bound);
  return VGX_OK;
}

// after a scan's kernel: ask for the counters without waiting for them -- but only once the
// unconfirmed bounds have eaten half of the pool's headroom: a copy between two kernels costs the
// stream a few microseconds, and at 16 scans of slack one look every ~8 scans is plenty
void request_readback(vgx_tsdf_layer L) {
  if (L->readback_inflight) return;
  int64_t pending = 0;
  for (auto& r : L->recent) pending += r.second;
  if (2 * pending < (int64_t)L->dev.max_blocks - L->known_blocks) return;
  vgx_ctx ctx = L->ctx;
  if (hipMemcpyAsync(L->h_stats, L->d_stats, sizeof(TsdfStats), hipMemcpyDeviceToHost, ctx->tsdf_stream) != hipSuccess ||
      hipEventRecord(L->readback_done, ctx->tsdf_stream) != hipSuccess) {
    (void)hipGetLastError();
    return;  // the synchronous path still works
  }
  L->readback_inflight = true;
  L->inflight_seq = L->scan_seq;
}

}  // namespace

}  // extern "C"

namespace vgx {
int tsdf_reserve_for_scan(vgx_tsdf_layer L, const float origin[3], float reach) { return reserve_for_scan(L, origin, reach); }
int64_t tsdf_last_scan_bound(vgx_tsdf_layer L) { return L->recent.empty() ? 0 : L->recent.back().second; }
void tsdf_request_readback(vgx_tsdf_layer L) { request_readback(L); }
}  // namespace vgx

extern "C" {

int vgx_tsdf_layer_create(vgx_ctx ctx, float voxel_size, int32_t vps, const int32_t lut_min[3],
                          const int32_t lut_dim[3], int32_t max_blocks, vgx_tsdf_layer* out) {
  if (!ctx || !out) return VGX_ERR_INVALID;
  *out = nullptr;
  std::lock_guard<std::mutex> lk(ctx->tsdf_mu);
  if (vps != 8 && vps != 16)
    return set_error(ctx, VGX_ERR_UNSUPPORTED, "vgx_tsdf_layer_create: voxels_per_side must be 8 or 16");
  if ((lut_min == nullptr) != (lut_dim == nullptr))
    return set_error(ctx, VGX_ERR_INVALID, "vgx_tsdf_layer_create: lut_min and lut_dim come together");
  if (lut_dim)
    for (int a = 0; a < 3; ++a)
      if (lut_dim[a] <= 0) return set_error(ctx, VGX_ERR_INVALID, "vgx_tsdf_layer_create: empty block box");
  if (!(voxel_size > 0)) return set_error(ctx, VGX_ERR_INVALID, "vgx_tsdf_layer_create: bad voxel_size");
  if (max_blocks <= 0) max_blocks = kDefaultPoolBlocks;
  VGX_HIP(ctx, hipSetDevice(ctx->device));
  vgx_tsdf_layer L = new (std::nothrow) vgx_tsdf_layer_s();
  if (!L) return set_error(ctx, VGX_ERR_NOMEM, "vgx_tsdf_layer_create: out of host memory");
  L->ctx = ctx;
  TsdfLayerDev& d = L->dev;
  d.vps = vps;
  d.vps_shift = vps == 16 ? 4 : 3;
  d.voxel_size = voxel_size;
  d.voxel_size_inv = 1.0f / voxel_size;
  int rc = VGX_OK;
  if (hipMalloc(&L->d_stats, sizeof(TsdfStats)) != hipSuccess ||
      hipHostMalloc((void**)&L->h_stats, sizeof(TsdfStats), hipHostMallocDefault) != hipSuccess ||
      hipEventCreateWithFlags(&L->readback_done, hipEventDisableTiming) != hipSuccess ||
      hipMemsetAsync(L->d_stats, 0, sizeof(TsdfStats), ctx->tsdf_stream) != hipSuccess)
    rc = set_error(ctx, VGX_ERR_NOMEM, "vgx_tsdf_layer_create: device allocation failed");
  if (rc == VGX_OK) {
    d.n_blocks = &L->d_stats->n_blocks;
    d.dropped = &L->d_stats->dropped;
    rc = grow_pool(L, max_blocks, 0);
  }
  // the box is only an initial reservation: scans re-box the table as they need (reserve_for_scan)
  if (rc == VGX_OK && lut_min) rc = rebox(L, lut_min, lut_dim);
  if (rc != VGX_OK) {
    vgx_tsdf_layer_destroy(L);
    return rc;
  }
  *out = L;
  return VGX_OK;
}

int vgx_tsdf_layer_destroy(vgx_tsdf_layer L) {
  if (!L) return VGX_ERR_INVALID;
  (void)hipSetDevice(L->ctx->device);
  (void)hipStreamSynchronize(L->ctx->tsdf_stream);
  TsdfLayerDev& d = L->dev;
  void* ptrs[] = {d.voxels, d.rgba, d.lut, d.block_index, L->d_stats};
  for (void* p : ptrs)
    if (p) (void)hipFree(p);
  if (L->h_stats) (void)hipHostFree(L->h_stats);
  if (L->readback_done) (void)hipEventDestroy(L->readback_done);
  delete L;
  return VGX_OK;
}

int vgx_tsdf_layer_stats(vgx_tsdf_layer L, int32_t* n_blocks, int64_t* dropped) {
  if (!L) return VGX_ERR_INVALID;
  std::lock_guard<std::mutex> lk(L->ctx->tsdf_mu);
  TsdfStats st{};
  int rc = read_stats_sync(L, &st);
  if (rc != VGX_OK) return rc;
  if (n_blocks) *n_blocks = st.n_blocks;
  if (dropped) *dropped = (int64_t)st.dropped;
  return VGX_OK;
}

int64_t vgx_tsdf_layer_growths(vgx_tsdf_layer L) { return L ? L->growths : -1; }

int vgx_tsdf_layer_clear_dropped(vgx_tsdf_layer L) {
  if (!L) return VGX_ERR_INVALID;
  vgx_ctx ctx = L->ctx;
  std::lock_guard<std::mutex> lk(ctx->tsdf_mu);
  VGX_HIP(ctx, hipSetDevice(ctx->device));
  VGX_HIP(ctx, hipStreamSynchronize(ctx->tsdf_stream));
  VGX_HIP(ctx, hipMemsetAsync(L->dev.dropped, 0, 8, ctx->tsdf_stream));
  VGX_HIP(ctx, hipStreamSynchronize(ctx->tsdf_stream));
  L->dropped_seen = 0;
  L->readback_inflight = false;  // a read-back in flight may still carry the old count
  return VGX_OK;
}

int vgx_tsdf_layer_reserve(vgx_tsdf_layer L, const float origin[3], float reach_m) {
  if (!L || !origin || !(reach_m >= 0)) return VGX_ERR_INVALID;
  std::lock_guard<std::mutex> lk(L->ctx->tsdf_mu);
  VGX_HIP(L->ctx, hipSetDevice(L->ctx->device));
  int rc = reserve_for_scan(L, origin, reach_m);
  // nothing was launched: the bound just booked must not count as a scan in flight
  if (rc == VGX_OK && !L->recent.empty()) L->recent.pop_back();
  return rc;
}

int vgx_tsdf_layer_download(vgx_tsdf_layer L, int32_t* block_index, float* distance, float* weight,
                            uint8_t* rgba) {
  if (!L) return VGX_ERR_INVALID;
  vgx_ctx ctx = L->ctx;
  int32_t nb = 0;
  int rc = vgx_tsdf_layer_stats(L, &nb, nullptr);
  if (rc != VGX_OK) return rc;
  if (nb == 0) return VGX_OK;
  std::lock_guard<std::mutex> lk(ctx->tsdf_mu);
  const size_t nvox = voxels_per_block(L->dev);
  if (block_index)
    VGX_HIP(ctx, hipMemcpy(block_index, L->dev.block_index, (size_t)nb * 12, hipMemcpyDeviceToHost));
  if (distance || weight) {
    std::vector<unsigned long long> v((size_t)nb * nvox);
    VGX_HIP(ctx, hipMemcpy(v.data(), L->dev.voxels, v.size() * 8, hipMemcpyDeviceToHost));
    for (size_t k = 0; k < v.size(); ++k) {
      uint32_t lo = (uint32_t)(v[k] & 0xffffffffull), hi = (uint32_t)(v[k] >> 32);
      if (distance) std::memcpy(&distance[k], &lo, 4);
      if (weight) std::memcpy(&weight[k], &hi, 4);
    }
  }
  if (rgba) VGX_HIP(ctx, hipMemcpy(rgba, L->dev.rgba, (size_t)nb * nvox * 4, hipMemcpyDeviceToHost));
  return VGX_OK;
}

// Replaces the layer's contents with host blocks (a voxblox::Layer<TsdfVoxel> handed over, e.g. a
// submap that already holds data when the GPU integrator takes over).
int vgx_tsdf_layer_upload(vgx_tsdf_layer L, int32_t n_blocks, const int32_t* block_index, const float* distance,
                          const float* weight, const uint8_t* rgba) {
  if (!L || n_blocks < 0) return VGX_ERR_INVALID;
  vgx_ctx ctx = L->ctx;
  std::lock_guard<std::mutex> lk(ctx->tsdf_mu);
  if (n_blocks > 0 && (!block_index || !distance || !weight))
    return set_error(ctx, VGX_ERR_INVALID, "vgx_tsdf_layer_upload: NULL arrays");
  VGX_HIP(ctx, hipSetDevice(ctx->device));
  VGX_HIP(ctx, hipStreamSynchronize(ctx->tsdf_stream));
  TsdfLayerDev& d = L->dev;
  const size_t vpb = voxels_per_block(d);
  // box of the uploaded blocks (with a little slack); the old table is dropped, not remapped
  int32_t mn[3] = {0, 0, 0}, mx[3] = {0, 0, 0};
  for (int b = 0; b < n_blocks; ++b)
    for (int a = 0; a < 3; ++a) {
      const int32_t v = block_index[3 * b + a];
      if (b == 0 || v < mn[a]) mn[a] = v;
      if (b == 0 || v > mx[a]) mx[a] = v;
    }
  if (d.lut) (void)hipFree(d.lut);
  d.lut = nullptr;
  L->lut_cells = 0;
  int rc = VGX_OK;
  if (n_blocks > 0) {
    int32_t dm[3];
    for (int a = 0; a < 3; ++a) {
      mn[a] -= 2;
      dm[a] = mx[a] - mn[a] + 3;
    }
    rc = rebox(L, mn, dm);
  }
  if (rc == VGX_OK) rc = grow_pool(L, std::max<int32_t>(d.max_blocks, n_blocks + kDefaultPoolBlocks), 0);
  if (rc != VGX_OK) return rc;
  if (n_blocks > 0) {
    const size_t nv = (size_t)n_blocks * vpb;
    DeviceScratch sd, sw;
    VGX_HIP(ctx, sd.alloc(nv * 4));
    VGX_HIP(ctx, sw.alloc(nv * 4));
    VGX_HIP(ctx, hipMemcpy(sd.p, distance, nv * 4, hipMemcpyHostToDevice));
    VGX_HIP(ctx, hipMemcpy(sw.p, weight, nv * 4, hipMemcpyHostToDevice));
    VGX_HIP(ctx, hipMemcpy(d.block_index, block_index, (size_t)n_blocks * 12, hipMemcpyHostToDevice));
    if (rgba) VGX_HIP(ctx, hipMemcpy(d.rgba, rgba, nv * 4, hipMemcpyHostToDevice));
    hipLaunchKernelGGL(tsdf_pack_kernel, dim3((unsigned)((nv + 255) / 256)), dim3(256), 0, ctx->tsdf_stream,
                       sd.as<float>(), sw.as<float>(), nv, d.voxels);
    hipLaunchKernelGGL(tsdf_lut_from_blocks_kernel, dim3((unsigned)((n_blocks + 255) / 256)), dim3(256), 0, ctx->tsdf_stream,
                       d.block_index, (int)n_blocks, make_int3(d.lut_min[0], d.lut_min[1], d.lut_min[2]),
                       make_int3(d.lut_dim[0], d.lut_dim[1], d.lut_dim[2]), d.lut);
    VGX_HIP(ctx, hipGetLastError());
  }
  TsdfStats st{};
  st.n_blocks = n_blocks;
  VGX_HIP(ctx, hipMemcpyAsync(L->d_stats, &st, sizeof(st), hipMemcpyHostToDevice, ctx->tsdf_stream));
  VGX_HIP(ctx, hipStreamSynchronize(ctx->tsdf_stream));
  L->known_blocks = n_blocks;
  L->known_seq = L->scan_seq;
  L->recent.clear();
  L->readback_inflight = false;
  L->dropped_seen = 0;
  return VGX_OK;
}

int vgx_submap_from_tsdf_layer(vgx_ctx ctx, vgx_tsdf_layer L, int32_t submap_id, vgx_submap* out) {
  if (!ctx || !L || !out || L->ctx != ctx) return VGX_ERR_INVALID;
  *out = nullptr;
  int32_t nb = 0;
  int rc = vgx_tsdf_layer_stats(L, &nb, nullptr);
  if (rc != VGX_OK) return rc;
  std::lock_guard<std::mutex> lk(ctx->tsdf_mu);
  vgx_submap sm = new (std::nothrow) vgx_submap_s();
  if (!sm) return set_error(ctx, VGX_ERR_NOMEM, "vgx_submap_from_tsdf_layer: out of host memory");
  const TsdfLayerDev& d = L->dev;
  sm->ctx = ctx;
  sm->id = submap_id;
  sm->vps = d.vps;
  sm->n_blocks = nb;
  sm->voxel_size = d.voxel_size;
  sm->voxel_size_inv = 1.0f / d.voxel_size;
  sm->block_size = (float)d.vps * d.voxel_size;
  sm->block_size_inv = 1.0f / sm->block_size;
  sm->block_index.resize(3 * (size_t)nb);
  if (nb > 0 && hipMemcpy(sm->block_index.data(), d.block_index, (size_t)nb * 12, hipMemcpyDeviceToHost) != hipSuccess)
    rc = set_error(ctx, VGX_ERR_HIP, "vgx_submap_from_tsdf_layer: block index download failed");
  if (rc == VGX_OK) rc = build_block_lut(sm);
  if (rc == VGX_OK && nb > 0) {
    const size_t nvox = (size_t)nb * d.vps * d.vps * d.vps;
    if (hipMalloc(&sm->d_block_index, (size_t)nb * 12) != hipSuccess ||
        hipMalloc(&sm->d_tsdf_distance, nvox * sizeof(float)) != hipSuccess ||
        hipMalloc(&sm->d_tsdf_weight, nvox * sizeof(float)) != hipSuccess) {
      rc = set_error(ctx, VGX_ERR_NOMEM, "vgx_submap_from_tsdf_layer: device allocation failed");
    } else {
      hipError_t e = hipMemcpyAsync(sm->d_block_index, d.block_index, (size_t)nb * 12,
                                    hipMemcpyDeviceToDevice, ctx->tsdf_stream);
      if (e == hipSuccess) {
        hipLaunchKernelGGL(tsdf_unpack_kernel, dim3((unsigned)((nvox + 255) / 256)), dim3(256), 0,
                           ctx->tsdf_stream, d.voxels, nvox, sm->d_tsdf_distance, sm->d_tsdf_weight);
        e = hipGetLastError();
      }
      // finishSubmap(): the TSDF side hands the layer's voxels to the registration side -- the submap's own kernels run
      // on the context's registration stream, behind an event on the TSDF stream
      if (e == hipSuccess) e = hipEventRecord(ctx->ev_handover, ctx->tsdf_stream);
      if (e != hipSuccess)
        rc = set_error(ctx, VGX_ERR_HIP, std::string("vgx_submap_from_tsdf_layer: ") + hipGetErrorString(e));
    }
    if (rc == VGX_OK) {
      std::lock_guard<std::mutex> reg(ctx->mu);  // (lock order: tsdf_mu, then mu)
      if (hipStreamWaitEvent(ctx->stream, ctx->ev_handover, 0) != hipSuccess)
        rc = set_error(ctx, VGX_ERR_HIP, "vgx_submap_from_tsdf_layer: hipStreamWaitEvent failed");
      if (rc == VGX_OK) rc = launch_brickify(sm, 0);
    }
  }
  if (rc != VGX_OK) {
    vgx_submap_destroy(sm);
    return rc;
  }
  *out = sm;
  return VGX_OK;
}

int vgx_tsdf_integrator_create(vgx_ctx ctx, const vgx_tsdf_config* cfg, vgx_tsdf_layer layer,
                               vgx_tsdf_integrator* out) {
  if (!ctx || !cfg || !out) return VGX_ERR_INVALID;
  *out = nullptr;
  std::lock_guard<std::mutex> lk(ctx->tsdf_mu);
  if (layer && layer->ctx != ctx)
    return set_error(ctx, VGX_ERR_INVALID, "vgx_tsdf_integrator_create: layer of another context");
  VGX_HIP(ctx, hipSetDevice(ctx->device));
  vgx_tsdf_integrator I = new (std::nothrow) vgx_tsdf_integrator_s();
  if (!I) return set_error(ctx, VGX_ERR_NOMEM, "vgx_tsdf_integrator_create: out of host memory");
  I->ctx = ctx;
  I->layer = layer;
  I->dev.cfg = *cfg;
  const size_t set_bytes = ((size_t)1 << kSetBits) * 8;
  bool ok = hipMalloc(&I->dev.start_set, set_bytes) == hipSuccess &&
            hipMalloc(&I->dev.observed_set, set_bytes) == hipSuccess &&
            hipMalloc(&I->dev.n_updates, 8 * kScanStatWords) == hipSuccess;
  const unsigned long long poison = ~0ull;
  if (ok)
    ok = hipMemset(I->dev.start_set, 0, set_bytes) == hipSuccess &&
         hipMemset(I->dev.observed_set, 0, set_bytes) == hipSuccess &&
         hipMemset(I->dev.n_updates, 0, 8 * kScanStatWords) == hipSuccess &&
         // the zero hash would look present in every zeroed slot (ApproxHashSet ctor)
         hipMemcpy(I->dev.start_set, &poison, 8, hipMemcpyHostToDevice) == hipSuccess &&
         hipMemcpy(I->dev.observed_set, &poison, 8, hipMemcpyHostToDevice) == hipSuccess;
  if (!ok) {
    vgx_tsdf_integrator_destroy(I);
    return set_error(ctx, VGX_ERR_NOMEM, "vgx_tsdf_integrator_create: device allocation failed");
  }
  ctx->tsdf_integrators.fetch_add(1);
  I->counted_on_ctx = true;
  *out = I;
  return VGX_OK;
}

int vgx_tsdf_integrator_destroy(vgx_tsdf_integrator I) {
  if (!I) return VGX_ERR_INVALID;
  if (I->counted_on_ctx) I->ctx->tsdf_integrators.fetch_sub(1);
  (void)hipSetDevice(I->ctx->device);
  (void)hipStreamSynchronize(I->ctx->tsdf_stream);
  void* ptrs[] = {I->dev.start_set, I->dev.observed_set, I->dev.n_updates, I->d_points, I->d_rgba, I->d_wg_stats, I->d_trace,
                  I->d_mkeys[0], I->d_mkeys[1], I->d_midx[0], I->d_midx[1], I->d_mstart, I->d_mcounters, I->d_msort,
                  I->d_mrank, I->d_gpg, I->d_gcolor, I->d_gflags, I->d_gcount, I->d_okey[0], I->d_okey[1], I->d_oidx[0],
                  I->d_oidx[1], I->d_osort};
  for (void* p : ptrs)
    if (p) (void)hipFree(p);
  for (int k = 0; k < 2; ++k) {
    if (I->h_stage[k]) (void)hipHostFree(I->h_stage[k]);
    if (I->stage_uploaded[k]) (void)hipEventDestroy(I->stage_uploaded[k]);
  }
  if (I->det) det_scratch_free(I->det);
  delete I;
  return VGX_OK;
}

int vgx_tsdf_integrator_set_cloud_width(vgx_tsdf_integrator I, int32_t width) {
  if (!I || width < 0) return VGX_ERR_INVALID;
  std::lock_guard<std::mutex> own(I->mu);
  I->cloud_width = width;
  return VGX_OK;
}

int vgx_tsdf_integrator_set_layer(vgx_tsdf_integrator I, vgx_tsdf_layer layer) {
  if (!I || !layer || layer->ctx != I->ctx) return VGX_ERR_INVALID;
  I->layer = layer;
  return VGX_OK;
}

// ApproxHashSet::resetApproxSet for one set
static int reset_set(vgx_ctx ctx, unsigned long long* set, unsigned long long* offset) {
  if (++(*offset) >= kFullResetThreshold) {
    const unsigned long long poison = ~0ull;
    VGX_HIP(ctx, hipMemsetAsync(set, 0, ((size_t)1 << kSetBits) * 8, ctx->tsdf_stream));
    *offset = 0;
    VGX_HIP(ctx, hipMemcpyAsync(set, &poison, 8, hipMemcpyHostToDevice, ctx->tsdf_stream));
    VGX_HIP(ctx, hipStreamSynchronize(ctx->tsdf_stream));
  }
  return VGX_OK;
}

// SortedThreadSafeIndex [recalled]: std::sort of (index, point_C.squaredNorm()) by ascending squared norm.
// The key is the bit pattern of the f32 Eigen reduction (x*x + y*y) + z*z (non-negative, so it orders like
// the float; NaN last); the radix sort is stable, so points at equal range stay in index order -- the tie
// rule include/voxgraph_amd.h and oracle/tsdf_oracle.c state (std::sort leaves it unspecified).
__global__ __launch_bounds__(256) void sorted_order_keys_kernel(const float* __restrict__ points_C, long long n,
                                                               uint32_t* __restrict__ key, uint32_t* __restrict__ idx) {
  const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const float x = points_C[3 * i], y = points_C[3 * i + 1], z = points_C[3 * i + 2];
  key[i] = __float_as_uint((x * x + y * y) + z * z);
  idx[i] = (uint32_t)i;
}

// The scan's visiting order: *order = nullptr for "mixed" (the kernels compute it), else the sorted table.
// The caller holds I->mu and ctx->tsdf_mu.
static int visiting_order(vgx_tsdf_integrator I, const void* d_points, int64_t n, const uint32_t** order) {
  vgx_ctx ctx = I->ctx;
  *order = nullptr;
  if (I->dev.cfg.integration_order == VGX_TSDF_ORDER_MIXED || n <= 0) return VGX_OK;
  if (I->dev.cfg.integration_order != VGX_TSDF_ORDER_SORTED)
    return set_error(ctx, VGX_ERR_INVALID, "vgx_tsdf_config.integration_order: neither VGX_TSDF_ORDER_MIXED nor VGX_TSDF_ORDER_SORTED");
  if (n >= (1ll << 32)) return set_error(ctx, VGX_ERR_UNSUPPORTED, "integration_order sorted: more than 2^32 points in a scan");
  if (n > I->order_cap) {
    VGX_HIP(ctx, hipStreamSynchronize(ctx->tsdf_stream));
    void* old[] = {I->d_okey[0], I->d_okey[1], I->d_oidx[0], I->d_oidx[1], I->d_osort};
    for (void* q : old)
      if (q) (void)hipFree(q);
    I->d_okey[0] = I->d_okey[1] = I->d_oidx[0] = I->d_oidx[1] = nullptr;
    I->d_osort = nullptr;
    I->order_cap = 0;
    for (int k = 0; k < 2; ++k) {
      VGX_HIP(ctx, hipMalloc(&I->d_okey[k], (size_t)n * 4));
      VGX_HIP(ctx, hipMalloc(&I->d_oidx[k], (size_t)n * 4));
    }
    size_t bytes = 0;
    VGX_HIP(ctx, stable_sort_pairs(nullptr, bytes, I->d_okey[0], I->d_okey[1], I->d_oidx[0], I->d_oidx[1], (size_t)n, 32,
                                   ctx->tsdf_stream));
    VGX_HIP(ctx, hipMalloc(&I->d_osort, std::max<size_t>(bytes, 16)));
    I->osort_bytes = bytes;
    I->order_cap = n;
  }
  hipLaunchKernelGGL(sorted_order_keys_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, ctx->tsdf_stream,
                     (const float*)d_points, (long long)n, I->d_okey[0], I->d_oidx[0]);
  VGX_HIP(ctx, hipGetLastError());
  size_t bytes = I->osort_bytes;
  VGX_HIP(ctx, stable_sort_pairs(I->d_osort, bytes, I->d_okey[0], I->d_okey[1], I->d_oidx[0], I->d_oidx[1], (size_t)n, 32,
                                 ctx->tsdf_stream));
  *order = I->d_oidx[1];
  return VGX_OK;
}

// integratePointCloud with the scan already in device memory; the caller holds I->mu.
static int integrate_locked(vgx_tsdf_integrator I, const float T[7], const void* d_points, const void* d_rgba,
                            int64_t n, int32_t freespace, int64_t* n_updates) {
  vgx_ctx ctx = I->ctx;
  std::lock_guard<std::mutex> lk(ctx->tsdf_mu);
  if (!I->layer) return set_error(ctx, VGX_ERR_INVALID, "vgx_tsdf_integrate: no layer set");
  VGX_HIP(ctx, hipSetDevice(ctx->device));
  // integratePointCloud: reset both approximate sets every clear_checks_every_n_frames
  if ((++I->reset_counter) >= I->dev.cfg.clear_checks_every_n_frames) {
    I->reset_counter = 0;
    int rc = reset_set(ctx, I->dev.start_set, &I->dev.start_offset);
    if (rc == VGX_OK) rc = reset_set(ctx, I->dev.observed_set, &I->dev.observed_offset);
    if (rc != VGX_OK) return rc;
  }
  if (n_updates) VGX_HIP(ctx, hipMemsetAsync(I->dev.n_updates, 0, 8 * kScanStatWords, ctx->tsdf_stream));
  if (n > 0) {
    // Every voxel a ray of this scan can touch lies within max_ray_length + truncation of the
    // sensor origin (a longer return is a clearing ray cut at max_ray_length, RayCaster [recalled]);
    // two voxels of slack cover the f32 rounding of the ray ends.
    const vgx_tsdf_config& c = I->dev.cfg;
    const float origin[3] = {T[4], T[5], T[6]};
    const float reach = c.max_ray_length_m + c.default_truncation_distance + 2.0f * I->layer->dev.voxel_size;
    int rc = reserve_for_scan(I->layer, origin, reach);
    if (rc != VGX_OK) return rc;
    if (c.deterministic) {
      // reproducible mode: the single-thread visiting order of the reference, resolved in parallel
      const uint32_t* order = nullptr;
      rc = visiting_order(I, d_points, n, &order);
      if (rc != VGX_OK) return rc;
      rc = det_integrate(I, T, d_points, d_rgba, n, freespace, order, n_updates);
      if (rc == VGX_OK) {
        request_readback(I->layer);
      } else {
        // The scan stopped half way: its start set is written, its observed set and its voxels are not.  With
        // clear_checks_every_n_frames > 1 a retried scan would find its own start cells present and cast
        // nothing.  Forget both sets -- what the next resetApproxSet would do -- and keep the error.
        const std::string why = vgx_last_error(ctx);
        I->reset_counter = 0;
        (void)reset_set(ctx, I->dev.start_set, &I->dev.start_offset);
        (void)reset_set(ctx, I->dev.observed_set, &I->dev.observed_offset);
        set_error(ctx, rc, why);
      }
      return rc;
    }
    // VGX_TSDF_KERNEL=v1: the one-thread-per-point kernel of rounds 1-4 (A/B runs: profiles/ab_tsdf_coop.sh)
    static const int kernel_version = [] {
      const char* e = getenv("VGX_TSDF_KERNEL");
      return (e && !strcmp(e, "v1")) ? 1 : 2;
    }();
    if (kernel_version == 2) {
      I->dev.wg_stats = nullptr;
      if (n_updates) {  // a counted scan: one row of statistics per workgroup
        const long long wgs = racing_scan_workgroups((long long)n, I->cloud_width);
        if (wgs > I->wg_stats_cap) {
          VGX_HIP(ctx, hipStreamSynchronize(ctx->tsdf_stream));
          if (I->d_wg_stats) (void)hipFree(I->d_wg_stats);
          I->d_wg_stats = nullptr;
          I->wg_stats_cap = 0;
          VGX_HIP(ctx, hipMalloc(&I->d_wg_stats, (size_t)wgs * kWgStatWords * 8));
          I->wg_stats_cap = wgs;
        }
        I->dev.wg_stats = I->d_wg_stats;
        I->wg_stats_rows = wgs;
      }
      VGX_HIP(ctx, (I->racing_launch ? I->racing_launch : launch_racing_scan)(ctx->tsdf_stream, I->layer->dev, I->dev, T, (const float*)d_points, (const uint32_t*)d_rgba,
                                      (long long)n, (int)freespace, n_updates != nullptr, I->cloud_width));
    } else {
      dim3 grid((unsigned)((n + 255) / 256)), block(256);
      static const bool pipelined = [] {
        const char* e = getenv("VGX_TSDF_PIPELINED");  // A/B switch (profiles/ab_tsdf.sh)
        return e ? atoi(e) != 0 : true;
      }();
      if (pipelined)
        hipLaunchKernelGGL(tsdf_integrate_kernel<true>, grid, block, 0, ctx->tsdf_stream, I->layer->dev, I->dev,
                           T[0], T[1], T[2], T[3], T[4], T[5], T[6], (const float*)d_points,
                           (const uint32_t*)d_rgba, (long long)n, (int)freespace);
      else
        hipLaunchKernelGGL(tsdf_integrate_kernel<false>, grid, block, 0, ctx->tsdf_stream, I->layer->dev, I->dev,
                           T[0], T[1], T[2], T[3], T[4], T[5], T[6], (const float*)d_points,
                           (const uint32_t*)d_rgba, (long long)n, (int)freespace);
    }
    VGX_HIP(ctx, hipGetLastError());
    request_readback(I->layer);
  }
  if (n_updates) {
    unsigned long long u = 0;
    VGX_HIP(ctx, hipMemcpyAsync(&u, I->dev.n_updates, 8, hipMemcpyDeviceToHost, ctx->tsdf_stream));
    VGX_HIP(ctx, hipStreamSynchronize(ctx->tsdf_stream));
    *n_updates = (int64_t)u;
  }
  return VGX_OK;
}

// MergedTsdfIntegrator::integratePointCloud with the scan already in device memory; the caller holds I->mu.
static int merged_integrate_body(vgx_tsdf_integrator I, const float T[7], const void* d_points, const void* d_rgba,
                                 int64_t n, int32_t freespace, int64_t* n_updates);
static int merged_integrate_locked(vgx_tsdf_integrator I, const float T[7], const void* d_points, const void* d_rgba,
                                   int64_t n, int32_t freespace, int64_t* n_updates) {
  vgx_ctx ctx = I->ctx;
  std::lock_guard<std::mutex> lk(ctx->tsdf_mu);
  const int rc = merged_integrate_body(I, T, d_points, d_rgba, n, freespace, n_updates);
  if (rc != VGX_OK && I->d_mcounters) {
    // a scan that stopped between its bundle and merge kernels would leave the "key corner" flag (counters[5]: set by
    // the bundle kernel, cleared by the merge kernel of the same scan) for the NEXT scan to be refused by
    const std::string why = vgx_last_error(ctx);
    (void)hipMemsetAsync(I->d_mcounters + 5, 0, 4, ctx->tsdf_stream);
    (void)hipGetLastError();
    set_error(ctx, rc, why);
  }
  return rc;
}

static int merged_integrate_body(vgx_tsdf_integrator I, const float T[7], const void* d_points, const void* d_rgba,
                                 int64_t n, int32_t freespace, int64_t* n_updates) {
  vgx_ctx ctx = I->ctx;
  if (!I->layer) return set_error(ctx, VGX_ERR_INVALID, "vgx_tsdf_integrate_merged: no layer set");
  if (n > (int64_t)1 << 31) return set_error(ctx, VGX_ERR_UNSUPPORTED, "vgx_tsdf_integrate_merged: more than 2^31 points");
  VGX_HIP(ctx, hipSetDevice(ctx->device));
  if (n_updates) VGX_HIP(ctx, hipMemsetAsync(I->dev.n_updates, 0, 8, ctx->tsdf_stream));
  if (n > 0) {
    const unsigned long long* keys_sorted = nullptr;
    const unsigned int* idx_sorted = nullptr;
    if (n > I->merged_cap) {
      VGX_HIP(ctx, hipStreamSynchronize(ctx->tsdf_stream));
      void* old[] = {I->d_mkeys[0], I->d_mkeys[1], I->d_midx[0], I->d_midx[1], I->d_mstart, I->d_msort,
                     I->d_mrank, I->d_gpg, I->d_gcolor, I->d_gflags, I->d_gcount};
      for (void* q : old)
        if (q) (void)hipFree(q);
      I->d_mkeys[0] = I->d_mkeys[1] = nullptr;
      I->d_midx[0] = I->d_midx[1] = nullptr;
      I->d_mstart = I->d_mrank = I->d_gcolor = I->d_gflags = I->d_gcount = nullptr;
      I->d_gpg = nullptr;
      I->d_msort = nullptr;
      I->merged_cap = 0;
      for (int k = 0; k < 2; ++k) {
        VGX_HIP(ctx, hipMalloc(&I->d_mkeys[k], (size_t)n * 8));
        VGX_HIP(ctx, hipMalloc(&I->d_midx[k], (size_t)n * 4));
      }
      VGX_HIP(ctx, hipMalloc(&I->d_mstart, (size_t)n * 4));
      VGX_HIP(ctx, hipMalloc(&I->d_gpg, (size_t)n * 16));
      VGX_HIP(ctx, hipMalloc(&I->d_gcolor, (size_t)n * 4));
      VGX_HIP(ctx, hipMalloc(&I->d_gflags, (size_t)n * 4));
      VGX_HIP(ctx, hipMalloc(&I->d_gcount, ((size_t)n + 1) * 4));
      if (!I->d_mcounters) {
        VGX_HIP(ctx, hipMalloc(&I->d_mcounters, 32));
        VGX_HIP(ctx, hipMemsetAsync(I->d_mcounters, 0, 32, ctx->tsdf_stream));  // ([5]: set by a scan, cleared by the same scan)
      }
      size_t bytes = 0;
      VGX_HIP(ctx, stable_sort_pairs_u64(nullptr, bytes, I->d_mkeys[0], I->d_mkeys[1], I->d_midx[0], I->d_midx[1], (size_t)n,
                                         ctx->tsdf_stream));
      VGX_HIP(ctx, hipMalloc(&I->d_msort, std::max<size_t>(bytes, 16)));
      I->msort_bytes = bytes;
      I->merged_cap = n;
    }
    const vgx_tsdf_config& c = I->dev.cfg;
    const float origin[3] = {T[4], T[5], T[6]};
    const float reach = c.max_ray_length_m + c.default_truncation_distance + 2.0f * I->layer->dev.voxel_size;
    int rc = reserve_for_scan(I->layer, origin, reach);
    if (rc != VGX_OK) return rc;
    const uint32_t* order = nullptr;  // the order a group's points are merged in
    rc = visiting_order(I, d_points, n, &order);
    if (rc != VGX_OK) return rc;
    unsigned long long* scan_ctr = nullptr;  // kCtr*: zeroed by the bundle kernel, read back once in det_merged_commit
    rc = det_counters(I, &scan_ctr);
    if (rc != VGX_OK) return rc;
    const dim3 grid((unsigned)((n + 255) / 256)), block(256);
    hipLaunchKernelGGL(merged_bundle_kernel, grid, block, 0, ctx->tsdf_stream, c, I->layer->dev.voxel_size_inv, T[0], T[1],
                       T[2], T[3], T[4], T[5], T[6], (const float*)d_points, (long long)n, order, (int)freespace,
                       I->d_mkeys[0], I->d_midx[0], I->d_mcounters, I->d_gcount, scan_ctr);
    VGX_HIP(ctx, hipGetLastError());
    size_t bytes = I->msort_bytes;
    // stable: equal keys keep the visiting order they were written in
    VGX_HIP(ctx, stable_sort_pairs_u64(I->d_msort, bytes, I->d_mkeys[0], I->d_mkeys[1], I->d_midx[0], I->d_midx[1], (size_t)n,
                                       ctx->tsdf_stream));
    keys_sorted = I->d_mkeys[1];
    idx_sorted = I->d_midx[1];
    {
      // group heads and their ranks in ONE launch (the prefix sum inside it: TileChain)
      const uint32_t head_tiles = (uint32_t)((n + 256 * kHeadsIpt - 1) / (256 * kHeadsIpt));
      TileChain chain;
      rc = det_next_chain(I, head_tiles, &chain);
      if (rc != VGX_OK) return rc;
      hipLaunchKernelGGL(merged_heads_kernel, dim3(head_tiles), block, 0, ctx->tsdf_stream, keys_sorted, (long long)n, chain, I->d_mstart,
                         I->d_mcounters);
      VGX_HIP(ctx, hipGetLastError());
    }
    // one L-lane sub-group per group, grid-stride (the number of groups stays on the device)
    // `lanes` per group (the merge itself is the same sequence of operations whatever the width: det_merged_commit
    // chooses it for the next scan from this scan's points per group)
    const int lanes = I->merged_lanes;
    const unsigned work_groups = (unsigned)std::min<long long>(((long long)n * lanes + 255) / 256, (long long)ctx->cu_count * 16);
#define VGX_LAUNCH_MERGE(L)                                                                                                        \
    hipLaunchKernelGGL(merged_merge_kernel<L>, dim3(work_groups), block, 0, ctx->tsdf_stream, c, I->layer->dev.voxel_size_inv, T[0],    \
                       T[1], T[2], T[3], T[4], T[5], T[6], (const float*)d_points, (const uint32_t*)d_rgba, keys_sorted, idx_sorted, \
                       I->d_mstart, I->d_mcounters, I->d_gpg, I->d_gcolor, I->d_gflags, I->d_gcount, I->d_mcounters + 5, scan_ctr)
    if (lanes == 4) VGX_LAUNCH_MERGE(4);
    else if (lanes == 8) VGX_LAUNCH_MERGE(8);
    else VGX_LAUNCH_MERGE(16);
#undef VGX_LAUNCH_MERGE
    VGX_HIP(ctx, hipGetLastError());
    // integrateRays.  Every ray crosses the sensor's own neighbourhood, so those voxels take one update
    // per group: thousands of rays contending for one compare-and-swap (measured: 30 ms per RGB-D
    // scan, most of it retries).  Instead the rays are written out, sorted by voxel and every voxel
    // applies its updates in group order -- surface groups in key order, then clearing groups: the
    // single thread's order (vgx_tsdf_det.hip).  The voxel VALUES are therefore the same on every run in
    // either mode; vgx_tsdf_config.deterministic additionally fixes the order blocks are allocated in.
    rc = det_merged_commit(I, T, (long long)n, I->d_gpg, I->d_gcolor, I->d_gflags, I->d_gcount, keys_sorted, I->d_mstart,
                           I->d_mcounters, n_updates);
    if (rc == VGX_OK) request_readback(I->layer);
    return rc;
  }
  if (n_updates) {  // an empty scan
    unsigned long long u = 0;
    VGX_HIP(ctx, hipMemcpyAsync(&u, I->dev.n_updates, 8, hipMemcpyDeviceToHost, ctx->tsdf_stream));
    VGX_HIP(ctx, hipStreamSynchronize(ctx->tsdf_stream));
    *n_updates = (int64_t)u;
  }
  return VGX_OK;
}

int vgx_tsdf_integrate_merged_device(vgx_tsdf_integrator I, const float T[7], const void* d_points,
                                     const void* d_rgba, int64_t n, int32_t freespace, int64_t* n_updates) {
  if (!I || !T || n < 0 || (n > 0 && !d_points)) return VGX_ERR_INVALID;
  std::lock_guard<std::mutex> own(I->mu);
  return merged_integrate_locked(I, T, d_points, d_rgba, n, freespace, n_updates);
}

static int stage_scan(vgx_tsdf_integrator I, const float* points, const uint8_t* rgba, int64_t n) {
  vgx_ctx ctx = I->ctx;
  std::lock_guard<std::mutex> lk(ctx->tsdf_mu);
  VGX_HIP(ctx, hipSetDevice(ctx->device));
  if (n > I->staging_cap) {
    VGX_HIP(ctx, hipStreamSynchronize(ctx->tsdf_stream));
    if (I->d_points) (void)hipFree(I->d_points);
    if (I->d_rgba) (void)hipFree(I->d_rgba);
    I->d_points = nullptr;
    I->d_rgba = nullptr;
    I->staging_cap = 0;
    VGX_HIP(ctx, hipMalloc(&I->d_points, (size_t)n * 12));
    VGX_HIP(ctx, hipMalloc(&I->d_rgba, (size_t)n * 4));
    I->staging_cap = n;
  }
  if (n > 0) {
    // Through pinned memory, two buffers in turn: the caller's arrays are read HERE (a host copy), the upload and the scan
    // run behind the call.  (Straight from pageable memory the runtime stages the copy itself and the call waits for it.)
    if (n > I->h_stage_cap) {
      VGX_HIP(ctx, hipStreamSynchronize(ctx->tsdf_stream));
      for (int k = 0; k < 2; ++k) {
        if (I->h_stage[k]) (void)hipHostFree(I->h_stage[k]);
        I->h_stage[k] = nullptr;
      }
      I->h_stage_cap = 0;
      bool ok = true;
      for (int k = 0; k < 2 && ok; ++k) {
        ok = hipHostMalloc((void**)&I->h_stage[k], (size_t)n * 16, hipHostMallocDefault) == hipSuccess;
        if (ok && !I->stage_uploaded[k]) ok = hipEventCreateWithFlags(&I->stage_uploaded[k], hipEventDisableTiming) == hipSuccess;
      }
      if (ok) {
        I->h_stage_cap = n;
      } else {  // no pinned memory to be had: the pageable path below
        (void)hipGetLastError();
        for (int k = 0; k < 2; ++k) {
          if (I->h_stage[k]) (void)hipHostFree(I->h_stage[k]);
          I->h_stage[k] = nullptr;
        }
      }
    }
    if (I->h_stage_cap >= n) {
      const int k = I->stage_turn;
      I->stage_turn ^= 1;
      VGX_HIP(ctx, hipEventSynchronize(I->stage_uploaded[k]));   // (the upload that last read this buffer: two scans ago)
      char* h = I->h_stage[k];
      std::memcpy(h, points, (size_t)n * 12);
      if (rgba) std::memcpy(h + (size_t)n * 12, rgba, (size_t)n * 4);
      VGX_HIP(ctx, hipMemcpyAsync(I->d_points, h, (size_t)n * 12, hipMemcpyHostToDevice, ctx->tsdf_stream));
      if (rgba) VGX_HIP(ctx, hipMemcpyAsync(I->d_rgba, h + (size_t)n * 12, (size_t)n * 4, hipMemcpyHostToDevice, ctx->tsdf_stream));
      VGX_HIP(ctx, hipEventRecord(I->stage_uploaded[k], ctx->tsdf_stream));
    } else {
      VGX_HIP(ctx, hipMemcpyAsync(I->d_points, points, (size_t)n * 12, hipMemcpyHostToDevice, ctx->tsdf_stream));
      if (rgba)
        VGX_HIP(ctx, hipMemcpyAsync(I->d_rgba, rgba, (size_t)n * 4, hipMemcpyHostToDevice, ctx->tsdf_stream));
      VGX_HIP(ctx, hipStreamSynchronize(ctx->tsdf_stream));   // the caller's arrays are his again when the call returns
    }
  }
  return VGX_OK;
}

int vgx_tsdf_integrate_merged(vgx_tsdf_integrator I, const float T[7], const float* points, const uint8_t* rgba,
                              int64_t n, int32_t freespace, int64_t* n_updates) {
  if (!I || !T || n < 0 || (n > 0 && !points)) return VGX_ERR_INVALID;
  std::lock_guard<std::mutex> own(I->mu);
  int rc = stage_scan(I, points, rgba, n);
  if (rc != VGX_OK) return rc;
  // (n_updates == NULL: nothing is counted and nothing is waited for -- the call returns with the scan queued)
  return merged_integrate_locked(I, T, I->d_points, rgba ? I->d_rgba : nullptr, n, freespace, n_updates);
}

int vgx_tsdf_integrate_device(vgx_tsdf_integrator I, const float T[7], const void* d_points,
                              const void* d_rgba, int64_t n, int32_t freespace, int64_t* n_updates) {
  if (!I || !T || n < 0 || (n > 0 && !d_points)) return VGX_ERR_INVALID;
  std::lock_guard<std::mutex> own(I->mu);
  return integrate_locked(I, T, d_points, d_rgba, n, freespace, n_updates);
}

int vgx_tsdf_integrate(vgx_tsdf_integrator I, const float T[7], const float* points, const uint8_t* rgba,
                       int64_t n, int32_t freespace, int64_t* n_updates) {
  if (!I || !T || n < 0 || (n > 0 && !points)) return VGX_ERR_INVALID;
  // the staging buffers are this scan's from the upload to the launch: two threads calling the
  // same integrator are serialised here, not interleaved between the two steps
  std::lock_guard<std::mutex> own(I->mu);
  int rc = stage_scan(I, points, rgba, n);
  if (rc != VGX_OK) return rc;
  // n_updates == NULL (what GpuFastTsdfIntegrator::integratePointCloud passes: voxblox's call returns nothing): an uncounted
  // scan, and the call returns with it QUEUED -- the points have been read, the layer is the device's and every reader of it
  // (download, finish, the next scan) is ordered behind the scan on the TSDF stream.  With a count the call waits for it.
  return integrate_locked(I, T, I->d_points, rgba ? I->d_rgba : nullptr, n, freespace, n_updates);
}

}  // extern "C"
