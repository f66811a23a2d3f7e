// Device-side VoxgraphSubmap::findRelevantVoxelIndices
// (voxgraph/src/frontend/submap_collection/voxgraph_submap.cpp:144-201):
// an order-preserving stream compaction of every TSDF voxel with
//   weight > min_voxel_weight && |distance| < max_voxel_distance   (VSM:177-178)
// into RegistrationPoint{voxel centre, ESDF-or-TSDF distance, TSDF weight}
// (VSM:180-197).  Pure scan: 8 B read per voxel (+4 B ESDF for kept voxels),
// 20 B written per kept voxel -- HBM-streaming bound.
#include <vector>

#include "vgx_internal.h"

#pragma clang fp contract(off)

namespace vgx {

__device__ __forceinline__ bool keep_voxel(float td, float tw, double min_w, double max_d) {
  return (double)tw > min_w && (double)fabsf(td) < max_d;
}

// pass 1: kept voxels and their f64 weight sum per block (fixed reduction tree)
template <int VPS>
__global__ __launch_bounds__(256) void extract_count_kernel(const float* __restrict__ tsdf_d,
                                                            const float* __restrict__ tsdf_w,
                                                            double min_w, double max_d,
                                                            int32_t* __restrict__ counts,
                                                            double* __restrict__ wsum) {
  constexpr int VOX = VPS * VPS * VPS;
  const size_t base = (size_t)blockIdx.x * VOX;
  int cnt = 0;
  double sw = 0.0;
  for (int v = threadIdx.x; v < VOX; v += 256) {
    float td = tsdf_d[base + v], tw = tsdf_w[base + v];
    if (keep_voxel(td, tw, min_w, max_d)) {
      ++cnt;
      sw += (double)tw;
    }
  }
#pragma unroll
  for (int off = 32; off >= 1; off >>= 1) {
    cnt += __shfl_xor(cnt, off, 64);
    sw += __shfl_xor(sw, off, 64);
  }
  __shared__ int s_cnt[4];
  __shared__ double s_sw[4];
  if ((threadIdx.x & 63) == 0) {
    s_cnt[threadIdx.x >> 6] = cnt;
    s_sw[threadIdx.x >> 6] = sw;
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    counts[blockIdx.x] = (s_cnt[0] + s_cnt[1]) + (s_cnt[2] + s_cnt[3]);
    wsum[blockIdx.x] = (s_sw[0] + s_sw[1]) + (s_sw[2] + s_sw[3]);
  }
}

// pass 2: write kept voxels of block b to [offsets[b], offsets[b+1]) in linear
// index order (the reference's inner loop order, VSM:170-172)
template <int VPS>
__global__ __launch_bounds__(256) void extract_write_kernel(
    const int32_t* __restrict__ block_index, const float* __restrict__ tsdf_d,
    const float* __restrict__ tsdf_w, const float* __restrict__ esdf_d, double min_w,
    double max_d, float voxel_size, float block_size, const int64_t* __restrict__ offsets,
    float4* __restrict__ xyzd, float* __restrict__ weight) {
  constexpr int VOX = VPS * VPS * VPS;
  const int b = blockIdx.x;
  const size_t base = (size_t)b * VOX;
  // Block::computeCoordinatesFromLinearIndex [recalled]: origin + (idx + 0.5) * voxel_size
  const float ox = (float)block_index[3 * b + 0] * block_size;
  const float oy = (float)block_index[3 * b + 1] * block_size;
  const float oz = (float)block_index[3 * b + 2] * block_size;
  __shared__ int s_wave[4];
  int64_t running = offsets[b];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  for (int round = 0; round < VOX; round += 256) {
    int v = round + (int)threadIdx.x;
    float td = 0.0f, tw = 0.0f;
    bool keep = false;
    if (v < VOX) {
      td = tsdf_d[base + v];
      tw = tsdf_w[base + v];
      keep = keep_voxel(td, tw, min_w, max_d);
    }
    unsigned long long mask = __ballot(keep);
    int before = __popcll(mask & ((1ull << lane) - 1ull));
    if (lane == 0) s_wave[wave] = __popcll(mask);
    __syncthreads();
    int wave_off = 0, total = 0;
#pragma unroll
    for (int w = 0; w < 4; ++w) {
      if (w < wave) wave_off += s_wave[w];
      total += s_wave[w];
    }
    if (keep) {
      int64_t at = running + wave_off + before;
      int ix = v % VPS, iy = (v / VPS) % VPS, iz = v / (VPS * VPS);
      float x = ox + ((float)ix + 0.5f) * voxel_size;
      float y = oy + ((float)iy + 0.5f) * voxel_size;
      float z = oz + ((float)iz + 0.5f) * voxel_size;
      float d = esdf_d ? esdf_d[base + v] : td;  // VSM:185-192
      xyzd[at] = make_float4(x, y, z, d);
      weight[at] = tw;  // VSM:195-197
    }
    running += total;
    __syncthreads();
  }
}

}  // namespace vgx

using namespace vgx;

extern "C" int vgx_submap_extract_voxel_points(vgx_submap sm, double min_voxel_weight,
                                               double max_voxel_distance,
                                               int32_t use_esdf_distance, int64_t* n_points_out) {
  if (!sm) return VGX_ERR_INVALID;
  vgx_ctx ctx = sm->ctx;
  std::lock_guard<std::mutex> lk(ctx->mu);
  if (sm->n_blocks > 0 && (!sm->d_tsdf_distance || !sm->d_tsdf_weight))
    return set_error(ctx, VGX_ERR_INVALID, "vgx_submap_extract_voxel_points: TSDF layer not resident");
  if (sm->n_blocks > 0 && use_esdf_distance && !sm->d_esdf_distance)
    return set_error(ctx, VGX_ERR_INVALID, "vgx_submap_extract_voxel_points: ESDF layer not resident");
  VGX_HIP(ctx, hipSetDevice(ctx->device));
  PointSet& ps = sm->points[VGX_POINTS_VOXELS];
  VGX_HIP(ctx, hipStreamSynchronize(ctx->stream));
  reset_point_set(ps);
  const int nb = sm->n_blocks;
  if (nb == 0) {
    ps.present = true;
    if (n_points_out) *n_points_out = 0;
    return VGX_OK;
  }
  DeviceScratch s_counts, s_wsum, s_offsets;
  VGX_HIP(ctx, s_counts.alloc((size_t)nb * sizeof(int32_t)));
  VGX_HIP(ctx, s_wsum.alloc((size_t)nb * sizeof(double)));
  VGX_HIP(ctx, s_offsets.alloc(((size_t)nb + 1) * sizeof(int64_t)));
  int32_t* d_counts = s_counts.as<int32_t>();
  double* d_wsum = s_wsum.as<double>();
  int64_t* d_offsets = s_offsets.as<int64_t>();
  if (sm->vps == 16)
    hipLaunchKernelGGL(extract_count_kernel<16>, dim3(nb), dim3(256), 0, ctx->stream,
                       sm->d_tsdf_distance, sm->d_tsdf_weight, min_voxel_weight,
                       max_voxel_distance, d_counts, d_wsum);
  else
    hipLaunchKernelGGL(extract_count_kernel<8>, dim3(nb), dim3(256), 0, ctx->stream,
                       sm->d_tsdf_distance, sm->d_tsdf_weight, min_voxel_weight,
                       max_voxel_distance, d_counts, d_wsum);
  std::vector<int32_t> counts((size_t)nb);
  std::vector<double> wsum((size_t)nb);
  std::vector<int64_t> offsets((size_t)nb + 1, 0);
  hipError_t e = hipMemcpyAsync(counts.data(), d_counts, (size_t)nb * sizeof(int32_t), hipMemcpyDeviceToHost, ctx->stream);
  if (e == hipSuccess) e = hipMemcpyAsync(wsum.data(), d_wsum, (size_t)nb * sizeof(double), hipMemcpyDeviceToHost, ctx->stream);
  if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
  int rc = VGX_OK;
  if (e != hipSuccess) rc = set_error(ctx, VGX_ERR_HIP, std::string("extract: ") + hipGetErrorString(e));
  double sum_w = 0;
  if (rc == VGX_OK) {
    // block order == the order the caller listed the blocks in (VSM:166)
    for (int b = 0; b < nb; ++b) {
      offsets[(size_t)b + 1] = offsets[(size_t)b] + counts[(size_t)b];
      sum_w += wsum[(size_t)b];
    }
    const int64_t n = offsets[(size_t)nb];
    if (n > INT32_MAX) rc = set_error(ctx, VGX_ERR_UNSUPPORTED, "extract: more than 2^31 points");
    ps.n = n;
    ps.sum_weight = sum_w;
    if (rc == VGX_OK && n > 0) {
      if (hipMalloc(&ps.d_xyzd, (size_t)n * sizeof(float4)) != hipSuccess ||
          hipMalloc(&ps.d_weight, (size_t)n * sizeof(float)) != hipSuccess)
        rc = set_error(ctx, VGX_ERR_NOMEM, "extract: device allocation failed");
    }
    if (rc == VGX_OK && n > 0) {
      e = hipMemcpyAsync(d_offsets, offsets.data(), ((size_t)nb + 1) * sizeof(int64_t), hipMemcpyHostToDevice, ctx->stream);
      if (e == hipSuccess) {
        const float* esdf = use_esdf_distance ? sm->d_esdf_distance : nullptr;
        if (sm->vps == 16)
          hipLaunchKernelGGL(extract_write_kernel<16>, dim3(nb), dim3(256), 0, ctx->stream,
                             sm->d_block_index, sm->d_tsdf_distance, sm->d_tsdf_weight, esdf,
                             min_voxel_weight, max_voxel_distance, sm->voxel_size, sm->block_size,
                             d_offsets, ps.d_xyzd, ps.d_weight);
        else
          hipLaunchKernelGGL(extract_write_kernel<8>, dim3(nb), dim3(256), 0, ctx->stream,
                             sm->d_block_index, sm->d_tsdf_distance, sm->d_tsdf_weight, esdf,
                             min_voxel_weight, max_voxel_distance, sm->voxel_size, sm->block_size,
                             d_offsets, ps.d_xyzd, ps.d_weight);
        e = hipGetLastError();
      }
      if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
      if (e != hipSuccess) rc = set_error(ctx, VGX_ERR_HIP, std::string("extract: ") + hipGetErrorString(e));
    }
  }
  if (rc == VGX_OK) rc = build_chunk_bounds(ctx, ps);
  if (rc == VGX_OK && n_points_out) *n_points_out = ps.n;
  // the set is offered to cost functions only once every allocation and kernel has succeeded
  if (rc == VGX_OK) ps.present = true; else reset_point_set(ps);
  return rc;
}
