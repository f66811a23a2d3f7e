// ESDF generation on the device: voxblox::EsdfIntegrator::updateFromTsdfLayerBatch
// [recalled], reached from voxgraph at
// voxgraph/src/frontend/submap_collection/voxgraph_submap.cpp:86 (generateEsdf()).
//
// voxblox propagates a wavefront through a bucketed priority queue -- inherently serial.
// The same recurrence
//     |d(v)| = min(|d(v)|, min over 26 neighbours n of same sign, |d(n)| < max: |d(n)| + step)
// is monotone, so it can be relaxed in any order to one fixed point: one workgroup per
// 16^3 block stages the block plus a one-voxel halo in LDS ((16+2)^3 floats = 23 KB),
// runs a few in-LDS sweeps (26 LDS reads per voxel per sweep, no global traffic), writes
// the block back and raises a flag if anything moved; the host repeats passes until the
// flag stays clear.  HBM traffic per pass: 4 B read (+ halo) + 4 B written per voxel.
#include <cmath>
#include <vector>

#include "vgx_internal.h"

#pragma clang fp contract(off)

namespace vgx {

struct TsdfLayerDevView;  // (vgx_tsdf.hip keeps its own definition)

__global__ __launch_bounds__(256) void esdf_init_kernel(const float* __restrict__ tsdf_d,
                                                        const float* __restrict__ tsdf_w,
                                                        size_t n, float min_weight,
                                                        float min_distance, float default_distance,
                                                        float* __restrict__ esdf_d,
                                                        uint8_t* __restrict__ esdf_obs) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  float w = tsdf_w[i], d = tsdf_d[i];
  if (w < min_weight) {  // unobserved in the TSDF: stays unobserved
    esdf_d[i] = 0.0f;
    esdf_obs[i] = 0;
    return;
  }
  esdf_obs[i] = 1;
  if (fabsf(d) < min_distance) {
    esdf_d[i] = d;  // fixed band
  } else {
    float sgn = (float)((d > 0.0f) - (d < 0.0f));
    esdf_d[i] = sgn * default_distance;
  }
}

constexpr int kLocalSweeps = 6;

template <int VPS>
__global__ __launch_bounds__(256) void esdf_relax_kernel(
    const int32_t* __restrict__ block_index, const int32_t* __restrict__ lut, int3 lut_min,
    int3 lut_dim, const float* __restrict__ tsdf_d, const uint8_t* __restrict__ esdf_obs,
    float* __restrict__ esdf_d, float voxel_size, float min_distance, float max_distance,
    int* __restrict__ changed) {
  constexpr int T = VPS + 2;
  constexpr int VOX = VPS * VPS * VPS;
  __shared__ float tile[T * T * T];
  __shared__ int s_changed;
  const int b = blockIdx.x;
  const int bx = block_index[3 * b] - lut_min.x, by = block_index[3 * b + 1] - lut_min.y,
            bz = block_index[3 * b + 2] - lut_min.z;
  if (threadIdx.x == 0) s_changed = 0;
  // stage block + halo; unobserved voxels and missing blocks are NaN (never a source)
  for (int c = threadIdx.x; c < T * T * T; c += 256) {
    int tx = c % T, ty = (c / T) % T, tz = c / (T * T);
    int vx = tx - 1, vy = ty - 1, vz = tz - 1;
    int ox = vx < 0 ? -1 : (vx >= VPS ? 1 : 0), oy = vy < 0 ? -1 : (vy >= VPS ? 1 : 0),
        oz = vz < 0 ? -1 : (vz >= VPS ? 1 : 0);
    int slot = b;
    if (ox | oy | oz) {
      int sx = bx + ox, sy = by + oy, sz = bz + oz;
      slot = -1;
      if ((unsigned)sx < (unsigned)lut_dim.x && (unsigned)sy < (unsigned)lut_dim.y &&
          (unsigned)sz < (unsigned)lut_dim.z)
        slot = lut[sx + lut_dim.x * (sy + lut_dim.y * sz)];
    }
    float v = __builtin_nanf("");
    if (slot >= 0) {
      size_t at = (size_t)slot * VOX + (size_t)((vx - ox * VPS) + VPS * ((vy - oy * VPS) + VPS * (vz - oz * VPS)));
      if (esdf_obs[at]) v = esdf_d[at];
    }
    tile[c] = v;
  }
  __syncthreads();
  const float s1 = voxel_size, s2 = sqrtf(2.0f) * voxel_size, s3 = sqrtf(3.0f) * voxel_size;
  bool any_change = false;
  for (int sweep = 0; sweep < kLocalSweeps; ++sweep) {
    bool sweep_change = false;
    for (int v = threadIdx.x; v < VOX; v += 256) {
      int x = v % VPS, y = (v / VPS) % VPS, z = v / (VPS * VPS);
      int c = (x + 1) + T * ((y + 1) + T * (z + 1));
      float cur = tile[c];
      if (!(cur == cur)) continue;                                   // unobserved
      if (fabsf(tsdf_d[(size_t)b * VOX + v]) < min_distance) continue;  // fixed band
      float mag = fabsf(cur);
      float best = mag;
#pragma unroll
      for (int dz = -1; dz <= 1; ++dz)
#pragma unroll
        for (int dy = -1; dy <= 1; ++dy)
#pragma unroll
          for (int dx = -1; dx <= 1; ++dx) {
            if (dx == 0 && dy == 0 && dz == 0) continue;
            float n = tile[c + dx + T * (dy + T * dz)];
            int nz = (dx != 0) + (dy != 0) + (dz != 0);
            float step = nz == 1 ? s1 : (nz == 2 ? s2 : s3);
            // same strict sign, source closer than max_distance (NaN fails every test)
            bool same = (cur > 0.0f && n > 0.0f) || (cur < 0.0f && n < 0.0f);
            float cand = fabsf(n) + step;
            if (same && fabsf(n) < max_distance && cand < best) best = cand;
          }
      if (best < mag) {
        tile[c] = cur > 0.0f ? best : -best;  // monotone: a stale read is only an upper bound
        sweep_change = true;
      }
    }
    any_change |= sweep_change;
    if (sweep_change) s_changed = 1;
    __syncthreads();
    int again = s_changed;
    __syncthreads();
    if (!again) break;
    if (threadIdx.x == 0) s_changed = 0;
    __syncthreads();
  }
  if (__syncthreads_or(any_change)) {
    for (int v = threadIdx.x; v < VOX; v += 256) {
      int x = v % VPS, y = (v / VPS) % VPS, z = v / (VPS * VPS);
      float val = tile[(x + 1) + T * ((y + 1) + T * (z + 1))];
      if (val == val) esdf_d[(size_t)b * VOX + v] = val;
    }
    if (threadIdx.x == 0) atomicExch(changed, 1);
  }
}

}  // namespace vgx

using namespace vgx;

extern "C" {

void vgx_esdf_config_default(vgx_esdf_config* c) {
  if (!c) return;
  c->max_distance_m = 2.0f;
  c->min_distance_m = 0.2f;
  c->default_distance_m = 2.0f;
  c->min_diff_m = 0.001f;
  c->min_weight = 1e-6f;
  c->num_buckets = 20;
}

int vgx_submap_generate_esdf(vgx_submap sm, const vgx_esdf_config* cfg_in, int32_t* sweeps_out) {
  if (!sm) return VGX_ERR_INVALID;
  vgx_ctx ctx = sm->ctx;
  std::lock_guard<std::mutex> lk(ctx->mu);
  vgx_esdf_config cfg;
  if (cfg_in) cfg = *cfg_in; else vgx_esdf_config_default(&cfg);
  if (sm->n_blocks > 0 && (!sm->d_tsdf_distance || !sm->d_tsdf_weight))
    return set_error(ctx, VGX_ERR_INVALID, "vgx_submap_generate_esdf: TSDF layer not resident");
  if (!(cfg.max_distance_m > 0) || !(cfg.min_distance_m >= 0) || !(cfg.default_distance_m > 0))
    return set_error(ctx, VGX_ERR_INVALID, "vgx_submap_generate_esdf: bad config");
  VGX_HIP(ctx, hipSetDevice(ctx->device));
  if (sweeps_out) *sweeps_out = 0;
  if (sm->n_blocks == 0) return VGX_OK;
  const size_t nvox = (size_t)sm->n_blocks * sm->vps * sm->vps * sm->vps;
  VGX_HIP(ctx, hipStreamSynchronize(ctx->stream));
  if (!sm->d_esdf_distance) VGX_HIP(ctx, hipMalloc(&sm->d_esdf_distance, nvox * sizeof(float)));
  if (!sm->d_esdf_observed) VGX_HIP(ctx, hipMalloc(&sm->d_esdf_observed, nvox));
  DeviceScratch s_changed;
  VGX_HIP(ctx, s_changed.alloc(sizeof(int)));
  int* d_changed = s_changed.as<int>();
  hipLaunchKernelGGL(esdf_init_kernel, dim3((unsigned)((nvox + 255) / 256)), dim3(256), 0, ctx->stream,
                     sm->d_tsdf_distance, sm->d_tsdf_weight, nvox, cfg.min_weight, cfg.min_distance_m,
                     cfg.default_distance_m, sm->d_esdf_distance, sm->d_esdf_observed);
  int3 mn = make_int3(sm->lut_min[0], sm->lut_min[1], sm->lut_min[2]);
  int3 dm = make_int3(sm->lut_dim[0], sm->lut_dim[1], sm->lut_dim[2]);
  int passes = 0, rc = VGX_OK;
  // the farthest a front travels is max_distance / voxel_size voxels; every pass
  // advances it by at least one voxel across block borders
  const int max_passes = 8 + (int)std::ceil(cfg.max_distance_m / sm->voxel_size) * 2;
  while (passes < max_passes) {
    hipError_t e = hipMemsetAsync(d_changed, 0, sizeof(int), ctx->stream);
    if (e == hipSuccess) {
      if (sm->vps == 16)
        hipLaunchKernelGGL(esdf_relax_kernel<16>, dim3(sm->n_blocks), dim3(256), 0, ctx->stream,
                           sm->d_block_index, sm->d_lut, mn, dm, sm->d_tsdf_distance,
                           sm->d_esdf_observed, sm->d_esdf_distance, sm->voxel_size,
                           cfg.min_distance_m, cfg.max_distance_m, d_changed);
      else
        hipLaunchKernelGGL(esdf_relax_kernel<8>, dim3(sm->n_blocks), dim3(256), 0, ctx->stream,
                           sm->d_block_index, sm->d_lut, mn, dm, sm->d_tsdf_distance,
                           sm->d_esdf_observed, sm->d_esdf_distance, sm->voxel_size,
                           cfg.min_distance_m, cfg.max_distance_m, d_changed);
      e = hipGetLastError();
    }
    int h_changed = 0;
    if (e == hipSuccess) e = hipMemcpyAsync(&h_changed, d_changed, sizeof(int), hipMemcpyDeviceToHost, ctx->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
    if (e != hipSuccess) {
      rc = set_error(ctx, VGX_ERR_HIP, std::string("vgx_submap_generate_esdf: ") + hipGetErrorString(e));
      break;
    }
    ++passes;
    if (!h_changed) break;
  }
  if (rc != VGX_OK) return rc;
  if (sweeps_out) *sweeps_out = passes;
  // (re)build the ESDF sampling grid
  if (sm->grid[1].d_bricks) {
    (void)hipFree(sm->grid[1].d_bricks);
    sm->grid[1].d_bricks = nullptr;
    sm->grid[1].present = false;
  }
  if (sm->grid[1].d_quad) {  // (made on demand from the bricks just dropped: made again when next asked for)
    (void)hipFree(sm->grid[1].d_quad);
    sm->grid[1].d_quad = nullptr;
  }
  return launch_brickify(sm, 1);
}

}  // extern "C"
