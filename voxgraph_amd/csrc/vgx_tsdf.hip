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
//     block lookup table (no hashing, no locks held across iterations).
// HBM-latency / atomic bound: 16 B per input point + 24 B per voxel update
// (SURVEY.md 8d); no MFMA.
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "vgx_internal.h"

#pragma clang fp contract(off)

namespace vgx {

struct TsdfLayerDev {
  unsigned long long* voxels;  // [max_blocks][vps^3] {distance (lo), weight (hi)}
  uint32_t* rgba;              // [max_blocks][vps^3]
  int32_t* lut;                // dense [dim.z][dim.y][dim.x]: slot, -1 free, -2 being allocated, -3 pool exhausted
  int32_t* block_index;        // [max_blocks][3]
  uint8_t* touched;            // [max_blocks]
  int32_t* n_blocks;           // allocation counter
  unsigned long long* dropped; // updates lost to box / pool limits
  int32_t lut_min[3], lut_dim[3];
  int32_t max_blocks, vps, vps_shift;
  float voxel_size, voxel_size_inv;
};

struct TsdfIntegratorDev {
  vgx_tsdf_config cfg;
  unsigned long long* start_set;     // [2^20]
  unsigned long long* observed_set;  // [2^20]
  unsigned long long start_offset, observed_offset;
  unsigned long long* n_updates;
};

constexpr unsigned kSetBits = 20;
constexpr unsigned kSetMask = (1u << kSetBits) - 1u;
constexpr unsigned long long kFullResetThreshold = 10000ull;

// ApproxHashSet::replaceHash with LongIndexHash [recalled]: true if the slot did
// not already hold this (hash + offset)
__device__ __forceinline__ bool approx_replace(unsigned long long* set, unsigned long long offset,
                                               int x, int y, int z) {
  unsigned int h = (unsigned int)x + (unsigned int)y * 17191u + (unsigned int)z * 295530481u;
  unsigned long long v = (unsigned long long)h + offset;
  unsigned long long old = atomicExch(&set[v & kSetMask], v);
  return old != v;
}

__device__ __forceinline__ float norm3(float x, float y, float z) {
  return sqrtf(x * x + y * y + z * z);
}

__device__ __forceinline__ int signum(float x) { return (x > 0.0f) - (x < 0.0f); }

// Layer::allocateBlockPtrByIndex without locks: returns the pool slot of block
// (bx,by,bz) or -1 when it lies outside the box / the pool is exhausted.
__device__ __forceinline__ int get_or_allocate_block(const TsdfLayerDev& L, int bx, int by, int bz) {
  int rx = bx - L.lut_min[0], ry = by - L.lut_min[1], rz = bz - L.lut_min[2];
  if ((unsigned)rx >= (unsigned)L.lut_dim[0] || (unsigned)ry >= (unsigned)L.lut_dim[1] ||
      (unsigned)rz >= (unsigned)L.lut_dim[2])
    return -1;
  int32_t* entry = &L.lut[rx + L.lut_dim[0] * (ry + L.lut_dim[1] * rz)];
  int slot = __hip_atomic_load(entry, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  // every lane stays in this loop; the lane that wins the CAS finishes the
  // allocation inside the same iteration, so nobody waits on a masked-off lane
  while (slot == -1 || slot == -2) {
    if (slot == -1 && atomicCAS(entry, -1, -2) == -1) {
      int s = atomicAdd(L.n_blocks, 1);
      if (s >= L.max_blocks) {
        atomicSub(L.n_blocks, 1);
        s = -3;
      } else {
        L.block_index[3 * s + 0] = bx;
        L.block_index[3 * s + 1] = by;
        L.block_index[3 * s + 2] = bz;
        __threadfence();
      }
      __hip_atomic_store(entry, s, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
      slot = s;
    } else {
      __builtin_amdgcn_s_sleep(1);
      slot = __hip_atomic_load(entry, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
  }
  return slot >= 0 ? slot : -1;
}

__device__ __forceinline__ unsigned long long pack_voxel(float d, float w) {
  return (unsigned long long)__float_as_uint(d) | ((unsigned long long)__float_as_uint(w) << 32);
}

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

__device__ __forceinline__ uint32_t blended_color(uint32_t oc, uint32_t color, float old_w, float w) {
  float total = old_w + w;
  float fw = old_w / total, sw = w / total;
  uint32_t nc = 0;
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    float a = (float)((oc >> (8 * k)) & 0xffu), b = (float)((color >> (8 * k)) & 0xffu);
    nc |= ((uint32_t)(uint8_t)roundf(a * fw + b * sw)) << (8 * k);
  }
  return nc;
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
  unsigned long long my_updates = 0, my_dropped = 0;
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
      int sx = (int)floorf(gx * sub_inv + 1e-6f), sy = (int)floorf(gy * sub_inv + 1e-6f),
          sz = (int)floorf(gz * sub_inv + 1e-6f);
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
          curr[a] = (int)floorf(ss[a] + 1e-6f);
          int end_index = (int)floorf(es[a] + 1e-6f);
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
            if (PIPELINED) pipeline_beat(L, c, color, s1, s2, s3);
            if (seen == v) ++collisions; else collisions = 0;
            if (collisions > c.max_consecutive_ray_collisions) break;
            int bx = vx >> shift, by = vy >> shift, bz = vz >> shift;  // floor division (vps = 2^shift)
            if (bx != last_b[0] || by != last_b[1] || bz != last_b[2]) {
              last_slot = get_or_allocate_block(L, bx, by, bz);
              last_b[0] = bx; last_b[1] = by; last_b[2] = bz;
              if (last_slot >= 0) L.touched[last_slot] = 1;
            }
            if (last_slot < 0) {
              ++my_dropped;
              continue;
            }
            size_t at = (size_t)last_slot * ((size_t)vps * vps * vps) +
                        (size_t)((vx & mask) + vps * ((vy & mask) + vps * (vz & mask)));
            if (PIPELINED)
              s1 = make_update(L, c, at, tx, ty, tz, gx, gy, gz, vx, vy, vz, weight);
            else
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
#pragma unroll
  for (int off = 32; off >= 1; off >>= 1) {
    my_updates += __shfl_xor(my_updates, off, 64);
    my_dropped += __shfl_xor(my_dropped, off, 64);
  }
  if ((threadIdx.x & 63) == 0) {
    if (my_updates) atomicAdd(I.n_updates, my_updates);
    if (my_dropped) atomicAdd(L.dropped, my_dropped);
  }
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

}  // namespace vgx

using namespace vgx;

struct vgx_tsdf_layer_s {
  vgx_ctx ctx = nullptr;
  TsdfLayerDev dev{};
  size_t lut_cells = 0;
};

struct vgx_tsdf_integrator_s {
  vgx_ctx ctx = nullptr;
  vgx_tsdf_layer layer = nullptr;
  TsdfIntegratorDev dev{};
  long long reset_counter = 0;
  float* d_points = nullptr;  // staging for host-pointer scans
  uint32_t* d_rgba = nullptr;
  long long staging_cap = 0;
};

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
}

int vgx_tsdf_layer_create(vgx_ctx ctx, float voxel_size, int32_t vps, const int32_t lut_min[3],
                          const int32_t lut_dim[3], int32_t max_blocks, vgx_tsdf_layer* out) {
  if (!ctx || !out || !lut_min || !lut_dim) return VGX_ERR_INVALID;
  *out = nullptr;
  std::lock_guard<std::mutex> lk(ctx->mu);
  if (vps != 8 && vps != 16)
    return set_error(ctx, VGX_ERR_UNSUPPORTED, "vgx_tsdf_layer_create: voxels_per_side must be 8 or 16");
  size_t cells = 1;
  for (int a = 0; a < 3; ++a) {
    if (lut_dim[a] <= 0) return set_error(ctx, VGX_ERR_INVALID, "vgx_tsdf_layer_create: empty block box");
    cells *= (size_t)lut_dim[a];
  }
  if (!(voxel_size > 0) || max_blocks <= 0 || cells > ((size_t)1 << 28))
    return set_error(ctx, VGX_ERR_INVALID, "vgx_tsdf_layer_create: bad voxel_size / max_blocks / box");
  VGX_HIP(ctx, hipSetDevice(ctx->device));
  vgx_tsdf_layer L = new (std::nothrow) vgx_tsdf_layer_s();
  if (!L) return set_error(ctx, VGX_ERR_NOMEM, "vgx_tsdf_layer_create: out of host memory");
  L->ctx = ctx;
  L->lut_cells = cells;
  TsdfLayerDev& d = L->dev;
  for (int a = 0; a < 3; ++a) {
    d.lut_min[a] = lut_min[a];
    d.lut_dim[a] = lut_dim[a];
  }
  d.max_blocks = max_blocks;
  d.vps = vps;
  d.vps_shift = vps == 16 ? 4 : 3;
  d.voxel_size = voxel_size;
  d.voxel_size_inv = 1.0f / voxel_size;
  const size_t nvox = (size_t)max_blocks * vps * vps * vps;
  bool ok = hipMalloc(&d.voxels, nvox * 8) == hipSuccess && hipMalloc(&d.rgba, nvox * 4) == hipSuccess &&
            hipMalloc(&d.lut, cells * 4) == hipSuccess &&
            hipMalloc(&d.block_index, (size_t)max_blocks * 12) == hipSuccess &&
            hipMalloc(&d.touched, (size_t)max_blocks) == hipSuccess &&
            hipMalloc(&d.n_blocks, 4) == hipSuccess && hipMalloc(&d.dropped, 8) == hipSuccess;
  if (ok)
    ok = hipMemsetAsync(d.voxels, 0, nvox * 8, ctx->stream) == hipSuccess &&
         hipMemsetAsync(d.rgba, 0, nvox * 4, ctx->stream) == hipSuccess &&
         hipMemsetAsync(d.lut, 0xff, cells * 4, ctx->stream) == hipSuccess &&
         hipMemsetAsync(d.touched, 0, (size_t)max_blocks, ctx->stream) == hipSuccess &&
         hipMemsetAsync(d.n_blocks, 0, 4, ctx->stream) == hipSuccess &&
         hipMemsetAsync(d.dropped, 0, 8, ctx->stream) == hipSuccess &&
         hipStreamSynchronize(ctx->stream) == hipSuccess;
  if (!ok) {
    vgx_tsdf_layer_destroy(L);
    return set_error(ctx, VGX_ERR_NOMEM, "vgx_tsdf_layer_create: device allocation failed");
  }
  *out = L;
  return VGX_OK;
}

int vgx_tsdf_layer_destroy(vgx_tsdf_layer L) {
  if (!L) return VGX_ERR_INVALID;
  (void)hipSetDevice(L->ctx->device);
  (void)hipStreamSynchronize(L->ctx->stream);
  TsdfLayerDev& d = L->dev;
  void* ptrs[] = {d.voxels, d.rgba, d.lut, d.block_index, d.touched, d.n_blocks, d.dropped};
  for (void* p : ptrs)
    if (p) (void)hipFree(p);
  delete L;
  return VGX_OK;
}

int vgx_tsdf_layer_stats(vgx_tsdf_layer L, int32_t* n_blocks, int64_t* dropped) {
  if (!L) return VGX_ERR_INVALID;
  vgx_ctx ctx = L->ctx;
  VGX_HIP(ctx, hipSetDevice(ctx->device));
  VGX_HIP(ctx, hipStreamSynchronize(ctx->stream));
  int32_t nb = 0;
  unsigned long long dr = 0;
  VGX_HIP(ctx, hipMemcpy(&nb, L->dev.n_blocks, 4, hipMemcpyDeviceToHost));
  VGX_HIP(ctx, hipMemcpy(&dr, L->dev.dropped, 8, hipMemcpyDeviceToHost));
  if (n_blocks) *n_blocks = nb;
  if (dropped) *dropped = (int64_t)dr;
  return VGX_OK;
}

int vgx_tsdf_layer_download(vgx_tsdf_layer L, int32_t* block_index, float* distance, float* weight,
                            uint8_t* rgba) {
  if (!L) return VGX_ERR_INVALID;
  vgx_ctx ctx = L->ctx;
  int32_t nb = 0;
  int rc = vgx_tsdf_layer_stats(L, &nb, nullptr);
  if (rc != VGX_OK) return rc;
  if (nb == 0) return VGX_OK;
  const size_t nvox = (size_t)L->dev.vps * L->dev.vps * L->dev.vps;
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

int vgx_submap_from_tsdf_layer(vgx_ctx ctx, vgx_tsdf_layer L, int32_t submap_id, vgx_submap* out) {
  if (!ctx || !L || !out || L->ctx != ctx) return VGX_ERR_INVALID;
  *out = nullptr;
  int32_t nb = 0;
  int rc = vgx_tsdf_layer_stats(L, &nb, nullptr);
  if (rc != VGX_OK) return rc;
  std::lock_guard<std::mutex> lk(ctx->mu);
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
                                    hipMemcpyDeviceToDevice, ctx->stream);
      if (e == hipSuccess) {
        hipLaunchKernelGGL(tsdf_unpack_kernel, dim3((unsigned)((nvox + 255) / 256)), dim3(256), 0,
                           ctx->stream, d.voxels, nvox, sm->d_tsdf_distance, sm->d_tsdf_weight);
        e = hipGetLastError();
      }
      if (e != hipSuccess)
        rc = set_error(ctx, VGX_ERR_HIP, std::string("vgx_submap_from_tsdf_layer: ") + hipGetErrorString(e));
    }
    if (rc == VGX_OK) rc = launch_brickify(sm, 0);
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
  std::lock_guard<std::mutex> lk(ctx->mu);
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
            hipMalloc(&I->dev.n_updates, 8) == hipSuccess;
  const unsigned long long poison = ~0ull;
  if (ok)
    ok = hipMemset(I->dev.start_set, 0, set_bytes) == hipSuccess &&
         hipMemset(I->dev.observed_set, 0, set_bytes) == hipSuccess &&
         hipMemset(I->dev.n_updates, 0, 8) == hipSuccess &&
         // the zero hash would look present in every zeroed slot (ApproxHashSet ctor)
         hipMemcpy(I->dev.start_set, &poison, 8, hipMemcpyHostToDevice) == hipSuccess &&
         hipMemcpy(I->dev.observed_set, &poison, 8, hipMemcpyHostToDevice) == hipSuccess;
  if (!ok) {
    vgx_tsdf_integrator_destroy(I);
    return set_error(ctx, VGX_ERR_NOMEM, "vgx_tsdf_integrator_create: device allocation failed");
  }
  *out = I;
  return VGX_OK;
}

int vgx_tsdf_integrator_destroy(vgx_tsdf_integrator I) {
  if (!I) return VGX_ERR_INVALID;
  (void)hipSetDevice(I->ctx->device);
  (void)hipStreamSynchronize(I->ctx->stream);
  void* ptrs[] = {I->dev.start_set, I->dev.observed_set, I->dev.n_updates, I->d_points, I->d_rgba};
  for (void* p : ptrs)
    if (p) (void)hipFree(p);
  delete I;
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
    VGX_HIP(ctx, hipMemsetAsync(set, 0, ((size_t)1 << kSetBits) * 8, ctx->stream));
    *offset = 0;
    VGX_HIP(ctx, hipMemcpyAsync(set, &poison, 8, hipMemcpyHostToDevice, ctx->stream));
    VGX_HIP(ctx, hipStreamSynchronize(ctx->stream));
  }
  return VGX_OK;
}

int vgx_tsdf_integrate_device(vgx_tsdf_integrator I, const float T[7], const void* d_points,
                              const void* d_rgba, int64_t n, int32_t freespace, int64_t* n_updates) {
  if (!I || !T || n < 0 || (n > 0 && !d_points)) return VGX_ERR_INVALID;
  vgx_ctx ctx = I->ctx;
  std::lock_guard<std::mutex> lk(ctx->mu);
  if (!I->layer) return set_error(ctx, VGX_ERR_INVALID, "vgx_tsdf_integrate: no layer set");
  VGX_HIP(ctx, hipSetDevice(ctx->device));
  // integratePointCloud: reset both approximate sets every clear_checks_every_n_frames
  if ((++I->reset_counter) >= I->dev.cfg.clear_checks_every_n_frames) {
    I->reset_counter = 0;
    int rc = reset_set(ctx, I->dev.start_set, &I->dev.start_offset);
    if (rc == VGX_OK) rc = reset_set(ctx, I->dev.observed_set, &I->dev.observed_offset);
    if (rc != VGX_OK) return rc;
  }
  if (n_updates) VGX_HIP(ctx, hipMemsetAsync(I->dev.n_updates, 0, 8, ctx->stream));
  if (n > 0) {
    dim3 grid((unsigned)((n + 255) / 256)), block(256);
    static const bool pipelined = [] {
      const char* e = getenv("VGX_TSDF_PIPELINED");  // A/B switch (profiles/ab_tsdf.sh)
      return e ? atoi(e) != 0 : true;
    }();
    if (pipelined)
      hipLaunchKernelGGL(tsdf_integrate_kernel<true>, grid, block, 0, ctx->stream, I->layer->dev, I->dev,
                         T[0], T[1], T[2], T[3], T[4], T[5], T[6], (const float*)d_points,
                         (const uint32_t*)d_rgba, (long long)n, (int)freespace);
    else
      hipLaunchKernelGGL(tsdf_integrate_kernel<false>, grid, block, 0, ctx->stream, I->layer->dev, I->dev,
                         T[0], T[1], T[2], T[3], T[4], T[5], T[6], (const float*)d_points,
                         (const uint32_t*)d_rgba, (long long)n, (int)freespace);
    VGX_HIP(ctx, hipGetLastError());
  }
  if (n_updates) {
    unsigned long long u = 0;
    VGX_HIP(ctx, hipMemcpyAsync(&u, I->dev.n_updates, 8, hipMemcpyDeviceToHost, ctx->stream));
    VGX_HIP(ctx, hipStreamSynchronize(ctx->stream));
    *n_updates = (int64_t)u;
  }
  return VGX_OK;
}

int vgx_tsdf_integrate(vgx_tsdf_integrator I, const float T[7], const float* points, const uint8_t* rgba,
                       int64_t n, int32_t freespace, int64_t* n_updates) {
  if (!I || !T || n < 0 || (n > 0 && !points)) return VGX_ERR_INVALID;
  vgx_ctx ctx = I->ctx;
  {
    std::lock_guard<std::mutex> lk(ctx->mu);
    VGX_HIP(ctx, hipSetDevice(ctx->device));
    if (n > I->staging_cap) {
      VGX_HIP(ctx, hipStreamSynchronize(ctx->stream));
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
      VGX_HIP(ctx, hipMemcpyAsync(I->d_points, points, (size_t)n * 12, hipMemcpyHostToDevice, ctx->stream));
      if (rgba)
        VGX_HIP(ctx, hipMemcpyAsync(I->d_rgba, rgba, (size_t)n * 4, hipMemcpyHostToDevice, ctx->stream));
    }
  }
  int64_t upd = 0;
  int rc = vgx_tsdf_integrate_device(I, T, I->d_points, rgba ? I->d_rgba : nullptr, n, freespace, &upd);
  if (n_updates) *n_updates = upd;
  return rc;
}

}  // extern "C"
