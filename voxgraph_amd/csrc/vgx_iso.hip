// Device-side VoxgraphSubmap::findIsosurfaceVertices
// (voxgraph/src/frontend/submap_collection/voxgraph_submap.cpp:203-243): the
// kIsosurfacePoints the shipped configuration registers with.  voxblox pieces restated
// [recalled]: MeshIntegrator (marching cubes on the dual cells, a cell is meshed iff its
// 8 corner voxels have weight > min_weight), MeshLayer::getConnectedMesh (one vertex per
// 0.5-voxel cell) and Interpolator::getVoxel (trilinear distance + weight).
//
// The vertex SET of marching cubes is "one zero crossing per sign-changing cell edge", so no
// triangle table is needed.  One workgroup per block stages sdf + observed flags of the
// block and a one-voxel halo in LDS; the 3 * 16^3 edges owned by the block are visited in
// canonical order (voxel linear index, axis) by four passes that recompute the cheap
// candidate test instead of storing candidates:
//   P0 count candidates            -> size of the dedup table
//   P1 dedup: cell -> smallest candidate id (open addressing, atomicCAS + atomicMin)
//   P2 count survivors (winner of its cell AND interpolable) per block -> host prefix sum
//   P3 write survivors in order (ballot prefix), 20 B per point
// All streaming; upload-time work, not on the per-iteration path.
#include <cmath>
#include <vector>

#include "vgx_internal.h"

#pragma clang fp contract(off)

namespace vgx {

constexpr unsigned long long kEmptyKey = ~0ull;

struct IsoParams {
  const int32_t* block_index;
  const int32_t* lut;
  int3 lut_min, lut_dim;
  const float* tsdf_d;
  const float* tsdf_w;
  float voxel_size, voxel_size_inv, block_size, block_size_inv;
  float min_weight;
  double threshold_inv;
  unsigned long long* keys;  // dedup table
  unsigned long long* ids;
  unsigned long long mask;
};

__device__ __forceinline__ unsigned long long hash64(unsigned long long k) {
  k ^= k >> 33;
  k *= 0xff51afd7ed558ccdull;
  k ^= k >> 33;
  k *= 0xc4ceb9fe1a85ec53ull;
  k ^= k >> 33;
  return k;
}

template <int VPS>
struct IsoTile {
  static constexpr int T = VPS + 2;
  float sdf[T * T * T];
  unsigned char ok[T * T * T];  // voxel exists and weight > min_weight
};

template <int VPS>
__device__ void iso_load_tile(const IsoParams& p, int b, IsoTile<VPS>& tile) {
  constexpr int T = VPS + 2;
  constexpr int VOX = VPS * VPS * VPS;
  const int bx = p.block_index[3 * b] - p.lut_min.x, by = p.block_index[3 * b + 1] - p.lut_min.y,
            bz = p.block_index[3 * b + 2] - p.lut_min.z;
  for (int c = threadIdx.x; c < T * T * T; c += 256) {
    int tx = c % T, ty = (c / T) % T, tz = c / (T * T);
    int vx = tx - 1, vy = ty - 1, vz = tz - 1;
    int ox = vx < 0 ? -1 : (vx >= VPS ? 1 : 0), oy = vy < 0 ? -1 : (vy >= VPS ? 1 : 0),
        oz = vz < 0 ? -1 : (vz >= VPS ? 1 : 0);
    int slot = b;
    if (ox | oy | oz) {
      int sx = bx + ox, sy = by + oy, sz = bz + oz;
      slot = -1;
      if ((unsigned)sx < (unsigned)p.lut_dim.x && (unsigned)sy < (unsigned)p.lut_dim.y &&
          (unsigned)sz < (unsigned)p.lut_dim.z)
        slot = p.lut[sx + p.lut_dim.x * (sy + p.lut_dim.y * sz)];
    }
    float s = 0.0f;
    unsigned char ok = 0;
    if (slot >= 0) {
      size_t at = (size_t)slot * VOX + (size_t)((vx - ox * VPS) + VPS * ((vy - oy * VPS) + VPS * (vz - oz * VPS)));
      s = p.tsdf_d[at];
      ok = p.tsdf_w[at] > p.min_weight;
    }
    tile.sdf[c] = s;
    tile.ok[c] = ok;
  }
}

// Edge e = lin * 3 + axis of block b: is it a marching-cubes vertex, and where?
template <int VPS>
__device__ __forceinline__ bool iso_candidate(const IsoParams& p, const IsoTile<VPS>& tile, int b,
                                              int e, float vert[3]) {
  constexpr int T = VPS + 2;
  const int lin = e / 3, axis = e - 3 * lin;
  const int v[3] = {lin % VPS, (lin / VPS) % VPS, lin / (VPS * VPS)};
  const int str[3] = {1, T, T * T};
  const int c0 = (v[0] + 1) + T * ((v[1] + 1) + T * (v[2] + 1));
  const int c1 = c0 + str[axis];
  if (!tile.ok[c0] || !tile.ok[c1]) return false;
  const float s0 = tile.sdf[c0], s1 = tile.sdf[c1];
  if ((s0 < 0.0f) == (s1 < 0.0f)) return false;  // no zero crossing on this edge
  // at least one of the 4 dual cells around the edge must be fully observed
  const int ob = (axis + 1) % 3, oc = (axis + 2) % 3;
  bool any_cell = false;
#pragma unroll
  for (int sb = 0; sb < 2; ++sb)
#pragma unroll
    for (int sc = 0; sc < 2; ++sc) {
      const int base = c0 - sb * str[ob] - sc * str[oc];
      bool all = true;
#pragma unroll
      for (int k = 0; k < 8; ++k)
        all = all && tile.ok[base + (k & 1) + T * (((k >> 1) & 1) + T * ((k >> 2) & 1))];
      any_cell = any_cell || all;
    }
  if (!any_cell) return false;
  // MarchingCubes::interpolateVertex, low -> high along the edge
  float p0[3];
#pragma unroll
  for (int a = 0; a < 3; ++a)
    p0[a] = (float)p.block_index[3 * b + a] * p.block_size + ((float)v[a] + 0.5f) * p.voxel_size;
  int nv = v[axis] + 1, nb = p.block_index[3 * b + axis];
  if (nv >= VPS) {
    nv -= VPS;
    nb++;
  }
  const float p1a = (float)nb * p.block_size + ((float)nv + 0.5f) * p.voxel_size;
  const float sdf_diff = s0 - s1;
  vert[0] = p0[0];
  vert[1] = p0[1];
  vert[2] = p0[2];
  if (fabsf(sdf_diff) >= 1e-6f) {
    const float t = s0 / sdf_diff;
    vert[axis] = p0[axis] + t * (p1a - p0[axis]);
  } else {
    vert[axis] = 0.5f * (p0[axis] + p1a);
  }
  return true;
}

// getConnectedMesh cell key: round(v / threshold) per axis, 21 signed bits each
__device__ __forceinline__ unsigned long long iso_cell_key(const IsoParams& p, const float vert[3]) {
  unsigned long long key = 0;
#pragma unroll
  for (int a = 0; a < 3; ++a) {
    long long c = (long long)round((double)vert[a] * p.threshold_inv);
    key |= ((unsigned long long)(c + (1ll << 20)) & 0x1fffffull) << (21 * a);
  }
  return key;
}

// Interpolator<TsdfVoxel>::getVoxelsAndQVector + interpVoxel on the raw layer
template <int VPS>
__device__ bool iso_interp(const IsoParams& p, const float pos[3], float& dist, float& wgt) {
  constexpr int VOX = VPS * VPS * VPS;
  int blk[3], vox[3];
  float dl[3];
#pragma unroll
  for (int a = 0; a < 3; ++a) {
    int b0 = (int)floorf(pos[a] * p.block_size_inv + 1e-6f);
    float origin = (float)b0 * p.block_size;
    int v = (int)floorf((pos[a] - origin) * p.voxel_size_inv + 1e-6f);
    v = min(max(v, 0), VPS - 1);
    float centre = origin + ((float)v + 0.5f) * p.voxel_size;
    if (a == 0) {
      // setIndexes: the block containing pos must exist (checked below through neighbours:
      // it is the block of one of the 8 neighbours)
    }
    if (pos[a] - centre < 0.0f) {
      v--;
      if (v < 0) {
        b0--;
        v += VPS;
      }
    }
    float origin2 = (float)b0 * p.block_size;
    dl[a] = (pos[a] - (origin2 + ((float)v + 0.5f) * p.voxel_size)) * p.voxel_size_inv;
    blk[a] = b0;
    vox[a] = v;
  }
  float d[8], w[8];
#pragma unroll
  for (int k = 0; k < 8; ++k) {
    int off[3] = {(k >> 2) & 1, (k >> 1) & 1, k & 1};
    int nb[3], nv[3];
#pragma unroll
    for (int a = 0; a < 3; ++a) {
      nb[a] = blk[a];
      nv[a] = vox[a] + off[a];
      if (nv[a] >= VPS) {
        nb[a]++;
        nv[a] -= VPS;
      }
    }
    int rx = nb[0] - p.lut_min.x, ry = nb[1] - p.lut_min.y, rz = nb[2] - p.lut_min.z;
    if ((unsigned)rx >= (unsigned)p.lut_dim.x || (unsigned)ry >= (unsigned)p.lut_dim.y ||
        (unsigned)rz >= (unsigned)p.lut_dim.z)
      return false;
    int slot = p.lut[rx + p.lut_dim.x * (ry + p.lut_dim.y * rz)];
    if (slot < 0) return false;
    size_t at = (size_t)slot * VOX + (size_t)(nv[0] + VPS * (nv[1] + VPS * nv[2]));
    d[k] = p.tsdf_d[at];
    w[k] = p.tsdf_w[at];
    if (!(w[k] > 0.0f)) return false;  // Interpolator<TsdfVoxel>::isVoxelValid
  }
  auto interp = [&](const float x[8]) {
    float c0 = x[0], c1 = -x[0] + x[4], c2 = -x[0] + x[2], c3 = -x[0] + x[1];
    float c4 = ((x[0] - x[2]) - x[4]) + x[6];
    float c5 = ((x[0] - x[1]) - x[2]) + x[3];
    float c6 = ((x[0] - x[1]) - x[4]) + x[5];
    float c7 = ((((((-x[0] + x[1]) + x[2]) - x[3]) + x[4]) - x[5]) - x[6]) + x[7];
    float q4 = dl[0] * dl[1], q5 = dl[1] * dl[2], q6 = dl[2] * dl[0], q7 = dl[0] * dl[1] * dl[2];
    return ((((((c0 + dl[0] * c1) + dl[1] * c2) + dl[2] * c3) + q4 * c4) + q5 * c5) + q6 * c6) + q7 * c7;
  };
  dist = interp(d);
  wgt = interp(w);
  return true;
}

// mode 0: count candidates; 1: dedup insert; 2: count survivors; 3: write survivors
template <int VPS, int MODE>
__global__ __launch_bounds__(256) void iso_pass_kernel(IsoParams p, int32_t* __restrict__ counts,
                                                      const int64_t* __restrict__ offsets,
                                                      float4* __restrict__ xyzd,
                                                      float* __restrict__ weight,
                                                      unsigned char* __restrict__ block_has_vertex,
                                                      const int32_t* __restrict__ block_list) {
  constexpr int EDGES = VPS * VPS * VPS * 3;
  __shared__ IsoTile<VPS> tile;
  __shared__ int s_wave[4];
  // passes 1-3 run on the blocks pass 0 found candidates in only (block_list; a block without a sign-changing edge of a
  // fully observed cell owns no vertex: most blocks of a submap)
  const int b = block_list ? block_list[blockIdx.x] : (int)blockIdx.x;
  iso_load_tile<VPS>(p, b, tile);
  __syncthreads();
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  int my_count = 0;
  int64_t running = MODE == 3 ? offsets[b] : 0;
  for (int round_ = 0; round_ < EDGES; round_ += 256) {
    const int e = round_ + (int)threadIdx.x;
    float vert[3] = {0, 0, 0};
    bool cand = e < EDGES && iso_candidate<VPS>(p, tile, b, e, vert);
    bool keep = cand;
    float dist = 0.0f, wgt = 0.0f;
    if (MODE >= 1 && cand) {
      const unsigned long long key = iso_cell_key(p, vert);
      const unsigned long long id = (unsigned long long)b * EDGES + (unsigned long long)e;
      unsigned long long h = hash64(key) & p.mask;
      if (MODE == 1) {
        while (true) {
          unsigned long long prev = atomicCAS(&p.keys[h], kEmptyKey, key);
          if (prev == kEmptyKey || prev == key) {
            atomicMin(&p.ids[h], id);
            break;
          }
          h = (h + 1) & p.mask;
        }
      } else {
        while (p.keys[h] != key) h = (h + 1) & p.mask;   // present: inserted in pass 1
        keep = p.ids[h] == id && iso_interp<VPS>(p, vert, dist, wgt);
      }
    }
    if (MODE == 0 || MODE == 2) my_count += keep ? 1 : 0;
    if (MODE == 3) {
      unsigned long long m = __ballot(keep);
      int before = __popcll(m & ((1ull << lane) - 1ull));
      if (lane == 0) s_wave[wave] = __popcll(m);
      __syncthreads();
      int wave_off = 0, total = 0;
#pragma unroll
      for (int w = 0; w < 4; ++w) {
        if (w < wave) wave_off += s_wave[w];
        total += s_wave[w];
      }
      if (keep) {
        int64_t at = running + wave_off + before;
        xyzd[at] = make_float4(vert[0], vert[1], vert[2], dist);
        weight[at] = wgt;
        // isosurface_blocks_: computeBlockIndexFromCoordinates(vertex) (VSM:237-240)
        int rx = (int)floorf(vert[0] * p.block_size_inv + 1e-6f) - p.lut_min.x;
        int ry = (int)floorf(vert[1] * p.block_size_inv + 1e-6f) - p.lut_min.y;
        int rz = (int)floorf(vert[2] * p.block_size_inv + 1e-6f) - p.lut_min.z;
        if ((unsigned)rx < (unsigned)p.lut_dim.x && (unsigned)ry < (unsigned)p.lut_dim.y &&
            (unsigned)rz < (unsigned)p.lut_dim.z) {
          int slot = p.lut[rx + p.lut_dim.x * (ry + p.lut_dim.y * rz)];
          if (slot >= 0) block_has_vertex[slot] = 1;
        }
      }
      running += total;
      __syncthreads();
    }
  }
  if (MODE == 0 || MODE == 2) {
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) my_count += __shfl_xor(my_count, off, 64);
    if (lane == 0) s_wave[wave] = my_count;
    __syncthreads();
    if (threadIdx.x == 0) counts[b] = (s_wave[0] + s_wave[1]) + (s_wave[2] + s_wave[3]);
  }
}

template <int MODE>
static void launch_iso(vgx_submap sm, const IsoParams& p, int32_t* counts, const int64_t* offsets,
                       float4* xyzd, float* weight, unsigned char* has, const int32_t* block_list = nullptr, int n_list = 0) {
  hipStream_t st = sm->ctx->stream;
  const int grid = block_list ? n_list : sm->n_blocks;
  if (grid <= 0) return;
  if (sm->vps == 16)
    hipLaunchKernelGGL((iso_pass_kernel<16, MODE>), dim3(grid), dim3(256), 0, st, p, counts,
                       offsets, xyzd, weight, has, block_list);
  else
    hipLaunchKernelGGL((iso_pass_kernel<8, MODE>), dim3(grid), dim3(256), 0, st, p, counts,
                       offsets, xyzd, weight, has, block_list);
}

}  // namespace vgx

using namespace vgx;

extern "C" int vgx_submap_extract_isosurface_points(vgx_submap sm, double min_voxel_weight,
                                                    int64_t* n_points_out) {
  if (!sm) return VGX_ERR_INVALID;
  vgx_ctx ctx = sm->ctx;
  std::lock_guard<std::mutex> lk(ctx->mu);
  if (sm->n_blocks > 0 && (!sm->d_tsdf_distance || !sm->d_tsdf_weight))
    return set_error(ctx, VGX_ERR_INVALID, "vgx_submap_extract_isosurface_points: TSDF layer not resident");
  VGX_HIP(ctx, hipSetDevice(ctx->device));
  VGX_HIP(ctx, hipStreamSynchronize(ctx->stream));
  PointSet& ps = sm->points[VGX_POINTS_ISOSURFACE];
  reset_point_set(ps);
  sm->isosurface_blocks.clear();
  if (sm->d_iso_block_index) {
    (void)hipFree(sm->d_iso_block_index);
    sm->d_iso_block_index = nullptr;
  }
  if (n_points_out) *n_points_out = 0;
  const int nb = sm->n_blocks;
  if (nb == 0) {
    ps.present = true;
    return VGX_OK;
  }

  IsoParams p{};
  p.block_index = sm->d_block_index;
  p.lut = sm->d_lut;
  p.lut_min = make_int3(sm->lut_min[0], sm->lut_min[1], sm->lut_min[2]);
  p.lut_dim = make_int3(sm->lut_dim[0], sm->lut_dim[1], sm->lut_dim[2]);
  p.tsdf_d = sm->d_tsdf_distance;
  p.tsdf_w = sm->d_tsdf_weight;
  p.voxel_size = sm->voxel_size;
  p.voxel_size_inv = sm->voxel_size_inv;
  p.block_size = sm->block_size;
  p.block_size_inv = sm->block_size_inv;
  p.min_weight = (float)min_voxel_weight;  // MeshIntegratorConfig::min_weight is a float (VSM:211-212)
  // getConnectedMesh(&mesh, 0.5 * voxel_size): FloatingPoint threshold, double inverse
  const float threshold = (float)(0.5 * (double)sm->voxel_size);
  p.threshold_inv = 1.0 / (double)threshold;

  int32_t* d_counts = nullptr;
  int64_t* d_offsets = nullptr;
  unsigned char* d_has = nullptr;
  std::vector<int32_t> counts((size_t)nb);
  std::vector<int64_t> offsets((size_t)nb + 1, 0);
  int rc = VGX_OK;
  auto fail = [&](hipError_t e) {
    rc = set_error(ctx, VGX_ERR_HIP, std::string("vgx_submap_extract_isosurface_points: ") + hipGetErrorString(e));
  };
  auto fetch_counts = [&]() {
    hipError_t e = hipGetLastError();
    if (e == hipSuccess) e = hipMemcpyAsync(counts.data(), d_counts, (size_t)nb * 4, hipMemcpyDeviceToHost, ctx->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
    if (e != hipSuccess) fail(e);
  };
  if (hipMalloc(&d_counts, (size_t)nb * 4) != hipSuccess || hipMalloc(&d_offsets, ((size_t)nb + 1) * 8) != hipSuccess ||
      hipMalloc(&d_has, (size_t)nb) != hipSuccess)
    rc = set_error(ctx, VGX_ERR_NOMEM, "vgx_submap_extract_isosurface_points: device allocation failed");
  if (rc == VGX_OK) {
    launch_iso<0>(sm, p, d_counts, nullptr, nullptr, nullptr, nullptr);
    fetch_counts();
  }
  int64_t candidates = 0;
  std::vector<int32_t> active;   // blocks with candidates, in block order
  int32_t* d_active = nullptr;
  if (rc == VGX_OK)
    for (int b = 0; b < nb; ++b) {
      candidates += counts[(size_t)b];
      if (counts[(size_t)b] > 0) active.push_back(b);
    }
  if (rc == VGX_OK && candidates > 0) {
    unsigned long long cap = 1024;
    while (cap < 2ull * (unsigned long long)candidates) cap <<= 1;
    p.mask = cap - 1;
    if (hipMalloc(&p.keys, cap * 8) != hipSuccess || hipMalloc(&p.ids, cap * 8) != hipSuccess) {
      rc = set_error(ctx, VGX_ERR_NOMEM, "vgx_submap_extract_isosurface_points: dedup table allocation failed");
    } else {
      hipError_t e = hipMemsetAsync(p.keys, 0xff, cap * 8, ctx->stream);
      if (e == hipSuccess) e = hipMemsetAsync(p.ids, 0xff, cap * 8, ctx->stream);
      if (e == hipSuccess) e = hipMemsetAsync(d_has, 0, (size_t)nb, ctx->stream);
      if (e == hipSuccess) e = hipMemsetAsync(d_counts, 0, (size_t)nb * 4, ctx->stream);   // (pass 2 writes the active blocks' only)
      if (e == hipSuccess) e = hipMalloc(&d_active, active.size() * 4);
      if (e == hipSuccess) e = hipMemcpyAsync(d_active, active.data(), active.size() * 4, hipMemcpyHostToDevice, ctx->stream);
      if (e != hipSuccess) fail(e);
    }
    if (rc == VGX_OK) {
      launch_iso<1>(sm, p, d_counts, nullptr, nullptr, nullptr, nullptr, d_active, (int)active.size());
      launch_iso<2>(sm, p, d_counts, nullptr, nullptr, nullptr, nullptr, d_active, (int)active.size());
      fetch_counts();
    }
    if (rc == VGX_OK) {
      for (int b = 0; b < nb; ++b) offsets[(size_t)b + 1] = offsets[(size_t)b] + counts[(size_t)b];
      const int64_t n = offsets[(size_t)nb];
      ps.n = n;
      if (n > 0) {
        if (hipMalloc(&ps.d_xyzd, (size_t)n * sizeof(float4)) != hipSuccess ||
            hipMalloc(&ps.d_weight, (size_t)n * sizeof(float)) != hipSuccess) {
          rc = set_error(ctx, VGX_ERR_NOMEM, "vgx_submap_extract_isosurface_points: point allocation failed");
        } else {
          hipError_t e = hipMemcpyAsync(d_offsets, offsets.data(), ((size_t)nb + 1) * 8, hipMemcpyHostToDevice, ctx->stream);
          if (e == hipSuccess) {
            launch_iso<3>(sm, p, d_counts, d_offsets, ps.d_xyzd, ps.d_weight, d_has, d_active, (int)active.size());
            e = hipGetLastError();
          }
          // sum of weights (RCF:124) and the isosurface block list
          std::vector<float> w((size_t)n);
          std::vector<unsigned char> has((size_t)nb);
          if (e == hipSuccess) e = hipMemcpyAsync(w.data(), ps.d_weight, (size_t)n * 4, hipMemcpyDeviceToHost, ctx->stream);
          if (e == hipSuccess) e = hipMemcpyAsync(has.data(), d_has, (size_t)nb, hipMemcpyDeviceToHost, ctx->stream);
          if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
          if (e != hipSuccess) {
            fail(e);
          } else {
            double sw = 0;
            for (int64_t i = 0; i < n; ++i) sw += (double)w[(size_t)i];
            ps.sum_weight = sw;
            for (int b = 0; b < nb; ++b)
              if (has[(size_t)b]) sm->isosurface_blocks.push_back(b);
          }
        }
      }
    }
  }
  if (p.keys) (void)hipFree(p.keys);
  if (p.ids) (void)hipFree(p.ids);
  if (d_counts) (void)hipFree(d_counts);
  if (d_active) (void)hipFree(d_active);
  if (d_offsets) (void)hipFree(d_offsets);
  if (d_has) (void)hipFree(d_has);
  if (rc == VGX_OK) rc = build_chunk_bounds(ctx, ps);
  if (rc == VGX_OK && !sm->isosurface_blocks.empty()) {
    std::vector<int32_t> ib(3 * sm->isosurface_blocks.size());
    for (size_t k = 0; k < sm->isosurface_blocks.size(); ++k)
      for (int a = 0; a < 3; ++a) ib[3 * k + a] = sm->block_index[3 * (size_t)sm->isosurface_blocks[k] + a];
    if (hipMalloc(&sm->d_iso_block_index, ib.size() * 4) != hipSuccess ||
        hipMemcpy(sm->d_iso_block_index, ib.data(), ib.size() * 4, hipMemcpyHostToDevice) != hipSuccess)
      rc = set_error(ctx, VGX_ERR_NOMEM, "vgx_submap_extract_isosurface_points: block list upload failed");
  }
  if (rc == VGX_OK && n_points_out) *n_points_out = ps.n;
  // the set is offered to cost functions only once every allocation and kernel has succeeded
  if (rc == VGX_OK) ps.present = true; else reset_point_set(ps);
  return rc;
}
