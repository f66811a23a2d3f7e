// Overlap detection, the callers' side of REG:
//   PoseGraphInterface::updateOverlappingSubmapList   pose_graph_interface.cpp:109-147
//   VoxgraphSubmap::overlapsWith                       voxgraph_submap.cpp:245-278
//   VoxgraphSubmap::getSubmapFrameSurfaceObb           voxgraph_submap.cpp:280-321
//   BoundingBox::getAabbFromObbAndPose                 bounding_box.cpp:28-42
// O(n^2) AABB rejects on the host (a few thousand float compares), then one launch over
// the surviving pairs: a workgroup per pair walks the isosurface blocks of submap i,
// carries each block centre into submap j's frame with the same f32 transform the
// registration uses, and probes j's dense block table.
#include <cmath>
#include <vector>

#include "vgx_internal.h"

#pragma clang fp contract(off)

namespace vgx {

struct OverlapSubmapDev {
  const int32_t* iso_block_index;  // [n_iso][3]
  int32_t n_iso;
  const int32_t* lut;
  int32_t lut_min[3], lut_dim[3];
  float block_size, block_size_inv;
};

struct OverlapPairDev {
  int32_t a, b;
  PosePack T;  // T_other__current (voxgraph_submap.cpp:261-262)
};

__global__ __launch_bounds__(256) void overlap_kernel(const OverlapSubmapDev* __restrict__ subs,
                                                      const OverlapPairDev* __restrict__ pairs,
                                                      int32_t* __restrict__ result) {
  const OverlapPairDev pr = pairs[blockIdx.x];
  const OverlapSubmapDev A = subs[pr.a], B = subs[pr.b];
  __shared__ int found;
  if (threadIdx.x == 0) found = 0;
  __syncthreads();
  for (int k = threadIdx.x; k < A.n_iso; k += 256) {
    if (found) break;
    // getCenterPointFromGridIndex(block_index, block_size): (idx + 0.5) * block_size
    float x = ((float)A.iso_block_index[3 * k] + 0.5f) * A.block_size;
    float y = ((float)A.iso_block_index[3 * k + 1] + 0.5f) * A.block_size;
    float z = ((float)A.iso_block_index[3 * k + 2] + 0.5f) * A.block_size;
    // T_other_submap__current_submap * t_current_submap__block (yaw-only, Eigen form)
    float uv0 = -(pr.T.qz * y), uv1 = pr.T.qz * x;
    uv0 += uv0;
    uv1 += uv1;
    float c0 = -(pr.T.qz * uv1), c1 = pr.T.qz * uv0;
    float px = (x + pr.T.qw * uv0 + c0) + pr.T.tx;
    float py = (y + pr.T.qw * uv1 + c1) + pr.T.ty;
    float pz = z + pr.T.tz;
    // getGridIndexFromPoint(p, block_size_inv) and Layer::hasBlock
    int bx = (int)floorf(px * B.block_size_inv + 1e-6f) - B.lut_min[0];
    int by = (int)floorf(py * B.block_size_inv + 1e-6f) - B.lut_min[1];
    int bz = (int)floorf(pz * B.block_size_inv + 1e-6f) - B.lut_min[2];
    if ((unsigned)bx < (unsigned)B.lut_dim[0] && (unsigned)by < (unsigned)B.lut_dim[1] &&
        (unsigned)bz < (unsigned)B.lut_dim[2] && B.lut[bx + B.lut_dim[0] * (by + B.lut_dim[1] * bz)] >= 0)
      found = 1;
  }
  __syncthreads();
  if (threadIdx.x == 0) result[blockIdx.x] = found;
}

// BoundingBox::getAabbFromObbAndPose for a yaw-only pose: the 8 OBB corners through
// voxblox::Transformation (f32 quaternion), component-wise min / max.
static void mission_aabb(const float obb_min[3], const float obb_max[3], const double pose[4],
                         float mn[3], float mx[3]) {
  // reuse the registration's pose arithmetic: T_mission__submap = exp(pose); with a zero
  // "reading" pose make_pose_pack returns exp(0)^-1 * exp(pose) = exp(pose)
  const double zero[4] = {0, 0, 0, 0};
  PosePack T;
  make_pose_pack(pose, zero, &T);
  for (int a = 0; a < 3; ++a) {
    mn[a] = INFINITY;
    mx[a] = -INFINITY;
  }
  for (unsigned i = 0; i < 8; ++i) {
    // getCornerCoordinates: bit set -> min, else max (bounding_box.cpp:12-25)
    float x = (i & 1) ? obb_min[0] : obb_max[0];
    float y = (i & 2) ? obb_min[1] : obb_max[1];
    float z = (i & 4) ? obb_min[2] : obb_max[2];
    float uv0 = -(T.qz * y), uv1 = T.qz * x;
    uv0 += uv0;
    uv1 += uv1;
    float c0 = -(T.qz * uv1), c1 = T.qz * uv0;
    float p[3] = {(x + T.qw * uv0 + c0) + T.tx, (y + T.qw * uv1 + c1) + T.ty, z + T.tz};
    for (int a = 0; a < 3; ++a) {
      mn[a] = std::fmin(mn[a], p[a]);
      mx[a] = std::fmax(mx[a], p[a]);
    }
  }
}

static int surface_obb(vgx_submap sm, float mn[3], float mx[3]) {
  const PointSet& ps = sm->points[VGX_POINTS_VOXELS];
  if (!ps.present || ps.n == 0)
    return set_error(sm->ctx, VGX_ERR_INVALID, "surface OBB: submap has no kVoxels registration points");
  const float half = 0.5f * sm->voxel_size;  // half_voxel_size (voxgraph_submap.cpp:296-297)
  for (int a = 0; a < 3; ++a) {
    mn[a] = ps.aabb_min[a] - half;
    mx[a] = ps.aabb_max[a] + half;
  }
  return VGX_OK;
}

}  // namespace vgx

using namespace vgx;

extern "C" {

int vgx_submap_surface_obb(vgx_submap sm, float mn[3], float mx[3]) {
  if (!sm || !mn || !mx) return VGX_ERR_INVALID;
  return surface_obb(sm, mn, mx);
}

int vgx_submap_mission_surface_aabb(vgx_submap sm, const double pose[4], float mn[3], float mx[3]) {
  if (!sm || !pose || !mn || !mx) return VGX_ERR_INVALID;
  float omn[3], omx[3];
  int rc = surface_obb(sm, omn, omx);
  if (rc != VGX_OK) return rc;
  mission_aabb(omn, omx, pose, mn, mx);
  return VGX_OK;
}

int vgx_find_overlapping_pairs(vgx_ctx ctx, int32_t n, const vgx_submap* submaps, const double* poses,
                               int32_t* pairs, int32_t max_pairs, int32_t* n_pairs) {
  if (!ctx || !n_pairs || n < 0 || (n > 0 && (!submaps || !poses)) || max_pairs < 0 ||
      (max_pairs > 0 && !pairs))
    return VGX_ERR_INVALID;
  *n_pairs = 0;
  std::lock_guard<std::mutex> lk(ctx->mu);
  VGX_HIP(ctx, hipSetDevice(ctx->device));
  std::vector<float> mn(3 * (size_t)n), mx(3 * (size_t)n);
  std::vector<OverlapSubmapDev> subs((size_t)n);
  for (int i = 0; i < n; ++i) {
    vgx_submap sm = submaps[i];
    if (!sm || sm->ctx != ctx) return set_error(ctx, VGX_ERR_INVALID, "vgx_find_overlapping_pairs: bad submap");
    if (!sm->points[VGX_POINTS_ISOSURFACE].present)
      return set_error(ctx, VGX_ERR_INVALID, "vgx_find_overlapping_pairs: submap has no isosurface points (not finished)");
    float omn[3], omx[3];
    int rc = surface_obb(sm, omn, omx);
    if (rc != VGX_OK) return rc;
    mission_aabb(omn, omx, poses + 4 * (size_t)i, &mn[3 * (size_t)i], &mx[3 * (size_t)i]);
    OverlapSubmapDev& d = subs[(size_t)i];
    d.iso_block_index = sm->d_iso_block_index;
    d.n_iso = (int32_t)sm->isosurface_blocks.size();
    d.lut = sm->d_lut;
    for (int a = 0; a < 3; ++a) {
      d.lut_min[a] = sm->lut_min[a];
      d.lut_dim[a] = sm->lut_dim[a];
    }
    d.block_size = sm->block_size;
    d.block_size_inv = sm->block_size_inv;
  }
  // stage 1: AABB separation along any axis (voxgraph_submap.cpp:248-256)
  std::vector<OverlapPairDev> cand;
  for (int i = 0; i < n; ++i)
    for (int j = i + 1; j < n; ++j) {
      bool sep = false;
      for (int a = 0; a < 3; ++a)
        sep = sep || mx[3 * (size_t)i + a] < mn[3 * (size_t)j + a] || mn[3 * (size_t)i + a] > mx[3 * (size_t)j + a];
      if (sep) continue;
      OverlapPairDev p;
      p.a = i;
      p.b = j;
      // T_other__current = other.getPose().inverse() * getPose(): "reference" = current (i)
      make_pose_pack(poses + 4 * (size_t)i, poses + 4 * (size_t)j, &p.T);
      cand.push_back(p);
    }
  if (cand.empty()) return VGX_OK;
  OverlapSubmapDev* d_subs = nullptr;
  OverlapPairDev* d_pairs = nullptr;
  int32_t* d_res = nullptr;
  std::vector<int32_t> res(cand.size());
  int rc = VGX_OK;
  hipError_t e = hipMalloc(&d_subs, subs.size() * sizeof(OverlapSubmapDev));
  if (e == hipSuccess) e = hipMalloc(&d_pairs, cand.size() * sizeof(OverlapPairDev));
  if (e == hipSuccess) e = hipMalloc(&d_res, cand.size() * sizeof(int32_t));
  if (e == hipSuccess) e = hipMemcpy(d_subs, subs.data(), subs.size() * sizeof(OverlapSubmapDev), hipMemcpyHostToDevice);
  if (e == hipSuccess) e = hipMemcpy(d_pairs, cand.data(), cand.size() * sizeof(OverlapPairDev), hipMemcpyHostToDevice);
  if (e == hipSuccess) {
    hipLaunchKernelGGL(overlap_kernel, dim3((unsigned)cand.size()), dim3(256), 0, ctx->stream, d_subs, d_pairs, d_res);
    e = hipGetLastError();
  }
  if (e == hipSuccess) e = hipMemcpyAsync(res.data(), d_res, res.size() * sizeof(int32_t), hipMemcpyDeviceToHost, ctx->stream);
  if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
  if (e != hipSuccess) rc = set_error(ctx, VGX_ERR_HIP, std::string("vgx_find_overlapping_pairs: ") + hipGetErrorString(e));
  if (d_subs) (void)hipFree(d_subs);
  if (d_pairs) (void)hipFree(d_pairs);
  if (d_res) (void)hipFree(d_res);
  if (rc != VGX_OK) return rc;
  int32_t k = 0;
  for (size_t c = 0; c < cand.size(); ++c)
    if (res[c]) {
      if (k < max_pairs) {
        pairs[2 * k] = cand[c].a;
        pairs[2 * k + 1] = cand[c].b;
      }
      ++k;
    }
  *n_pairs = k;
  if (k > max_pairs) return set_error(ctx, VGX_ERR_INVALID, "vgx_find_overlapping_pairs: pairs buffer too small");
  return VGX_OK;
}

}  // extern "C"
