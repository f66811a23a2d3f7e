// Context, submap upload and the apron-brick re-layout kernel.
#include <algorithm>
#include <atomic>
#include <cmath>
#include <cstring>
#include <numeric>

#include <cstdlib>

#include "vgx_internal.h"

namespace vgx {

static std::mutex g_err_mu;
static std::string g_global_error = "no error";

void set_global_error(const std::string& msg) {
  std::lock_guard<std::mutex> lk(g_err_mu);
  g_global_error = msg;
}

// Errors are recorded from any thread (Ceres evaluates on 4, pose_graph.cpp:96), with or without
// the context lock held, so the message has its own small lock.
int set_error(vgx_ctx ctx, int code, const std::string& msg) {
  if (ctx) {
    std::lock_guard<std::mutex> lk(ctx->err_mu);
    ctx->last_error = msg;
  } else {
    set_global_error(msg);
  }
  return code;
}

// ---------------------------------------------------------------------------
// brickify: voxblox block layout -> apron bricks with NaN validity
// ---------------------------------------------------------------------------
// One workgroup per block.  mode 0: valid iff weight > 0 (Interpolator<TsdfVoxel>
// ::isVoxelValid), mode 1: valid iff observed (Interpolator<EsdfVoxel>) [recalled].
template <int VPS, int LAYOUT>
__global__ __launch_bounds__(256) void brickify_kernel(
    const int32_t* __restrict__ block_index, const int32_t* __restrict__ lut,
    int3 lut_min, int3 lut_dim, const float* __restrict__ distance,
    const float* __restrict__ weight, const uint8_t* __restrict__ observed,
    float* __restrict__ bricks) {
  constexpr int CELLS = BrickLayout<VPS, LAYOUT>::cells;
  constexpr int VOX = VPS * VPS * VPS;
  const int b = blockIdx.x;
  const int bx = block_index[3 * b + 0] - lut_min.x;
  const int by = block_index[3 * b + 1] - lut_min.y;
  const int bz = block_index[3 * b + 2] - lut_min.z;
  float* out = bricks + (size_t)b * CELLS;
  for (int cell = threadIdx.x; cell < CELLS; cell += blockDim.x) {
    int cx, cy, cz;
    if (!BrickLayout<VPS, LAYOUT>::decode(cell, cx, cy, cz)) {
      out[cell] = 0.0f;  // padding, never read
      continue;
    }
    int ox = cx == VPS, oy = cy == VPS, oz = cz == VPS;
    int sx = bx + ox, sy = by + oy, sz = bz + oz;
    int slot = b;
    if (ox | oy | oz) {
      slot = -1;
      if (sx < lut_dim.x && sy < lut_dim.y && sz < lut_dim.z)
        slot = lut[sx + lut_dim.x * (sy + lut_dim.y * sz)];
    }
    float v = __builtin_nanf("");
    if (slot >= 0) {
      int vx = ox ? 0 : cx, vy = oy ? 0 : cy, vz = oz ? 0 : cz;
      size_t at = (size_t)slot * VOX + vx + VPS * (vy + VPS * vz);
      bool ok = weight ? (weight[at] > 0.0f) : (observed[at] != 0);
      if (ok) v = distance[at];
    }
    out[cell] = v;
  }
}

int launch_brickify(vgx_submap sm, int which) {
  vgx_ctx ctx = sm->ctx;
  const float* dist = which == 0 ? sm->d_tsdf_distance : sm->d_esdf_distance;
  const float* w = which == 0 ? sm->d_tsdf_weight : nullptr;
  const uint8_t* obs = which == 0 ? nullptr : sm->d_esdf_observed;
  const int layout = ctx->brick_layout;
  const size_t cells = brick_cells(sm->vps, layout);
  // the fused REG kernel addresses a brick cell with a 32-bit `slot * cells + offset` (vgx_reg.hip,
  // reg_eval_reduce_lean_kernel): a submap whose bricks do not fit that range is refused here, at upload
  // time (apron bricks, vps 16: 874 K blocks = a 1500^3-voxel dense cube; quad bricks: 246 K blocks)
  if ((unsigned long long)sm->n_blocks * cells >= (1ull << 32))
    return set_error(ctx, VGX_ERR_UNSUPPORTED,
                     "submap too large for 32-bit brick addressing: " + std::to_string(sm->n_blocks) + " blocks x " +
                         std::to_string(cells) + " cells per brick");
  size_t bytes = (size_t)sm->n_blocks * cells * sizeof(float);
  VGX_HIP(ctx, hipMalloc(&sm->grid[which].d_bricks, bytes));
  sm->grid[which].layout = layout;
  int3 mn = make_int3(sm->lut_min[0], sm->lut_min[1], sm->lut_min[2]);
  int3 dm = make_int3(sm->lut_dim[0], sm->lut_dim[1], sm->lut_dim[2]);
#define VGX_BRICKIFY(VPS, LAYOUT)                                                                       \
  hipLaunchKernelGGL((brickify_kernel<VPS, LAYOUT>), dim3(sm->n_blocks), dim3(256), 0, ctx->stream,    \
                     sm->d_block_index, sm->d_lut, mn, dm, dist, w, obs, sm->grid[which].d_bricks)
  if (sm->vps == 16) {
    if (layout == 0) VGX_BRICKIFY(16, 0); else if (layout == 1) VGX_BRICKIFY(16, 1); else VGX_BRICKIFY(16, 2);
  } else {
    if (layout == 0) VGX_BRICKIFY(8, 0); else if (layout == 1) VGX_BRICKIFY(8, 1); else VGX_BRICKIFY(8, 2);
  }
#undef VGX_BRICKIFY
  VGX_HIP(ctx, hipGetLastError());
  sm->grid[which].present = true;
  return VGX_OK;
}

// apron bricks -> quad bricks (same values: a copy, cell by cell)
template <int VPS>
__global__ __launch_bounds__(256) void requad_kernel(const float* __restrict__ apron, float* __restrict__ quad) {
  constexpr int CA = BrickLayout<VPS, 0>::cells, CQ = BrickLayout<VPS, 1>::cells, B = VPS + 1;
  const float* in = apron + (size_t)blockIdx.x * CA;
  float* out = quad + (size_t)blockIdx.x * CQ;
  for (int i = threadIdx.x; i < CQ; i += blockDim.x) {
    int cx, cy, cz;
    BrickLayout<VPS, 1>::decode(i, cx, cy, cz);
    out[i] = in[cx + B * (cy + B * cz)];
  }
}

// Bounding sphere of every kChunkPoints consecutive registration points (one
// wavefront per chunk).  The fused REG pass tests a chunk's sphere against the reading
// grid's box before it requests the chunk's points at all.
__global__ __launch_bounds__(64) void chunk_bounds_kernel(const float4* __restrict__ xyzd, long long n,
                                                         float4* __restrict__ bounds,
                                                         float* __restrict__ chunk_minmax) {
  const long long first = (long long)blockIdx.x * kChunkPoints;
  float mn[3] = {INFINITY, INFINITY, INFINITY}, mx[3] = {-INFINITY, -INFINITY, -INFINITY};
  for (long long i = first + threadIdx.x; i < first + kChunkPoints && i < n; i += 64) {
    float4 p = xyzd[i];
    mn[0] = fminf(mn[0], p.x); mx[0] = fmaxf(mx[0], p.x);
    mn[1] = fminf(mn[1], p.y); mx[1] = fmaxf(mx[1], p.y);
    mn[2] = fminf(mn[2], p.z); mx[2] = fmaxf(mx[2], p.z);
  }
#pragma unroll
  for (int a = 0; a < 3; ++a)
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) {
      mn[a] = fminf(mn[a], __shfl_xor(mn[a], off, 64));
      mx[a] = fmaxf(mx[a], __shfl_xor(mx[a], off, 64));
    }
  if (threadIdx.x == 0) {
    float hx = 0.5f * (mx[0] - mn[0]), hy = 0.5f * (mx[1] - mn[1]), hz = 0.5f * (mx[2] - mn[2]);
    // radius rounded up generously: the test it feeds is conservative anyway
    float r = sqrtf(hx * hx + hy * hy + hz * hz) * 1.0001f + 1e-4f;
    bounds[blockIdx.x] = make_float4(0.5f * (mx[0] + mn[0]), 0.5f * (mx[1] + mn[1]),
                                     0.5f * (mx[2] + mn[2]), r);
#pragma unroll
    for (int a = 0; a < 3; ++a) {
      chunk_minmax[6 * blockIdx.x + a] = mn[a];
      chunk_minmax[6 * blockIdx.x + 3 + a] = mx[a];
    }
  }
}

int build_chunk_bounds(vgx_ctx ctx, PointSet& ps) {
  if (ps.d_chunk_bounds) {
    (void)hipFree(ps.d_chunk_bounds);
    ps.d_chunk_bounds = nullptr;
  }
  if (ps.n <= 0) return VGX_OK;
  const long long chunks = (ps.n + kChunkPoints - 1) / kChunkPoints;
  VGX_HIP(ctx, hipMalloc(&ps.d_chunk_bounds, (size_t)chunks * sizeof(float4)));
  DeviceScratch s_minmax;
  VGX_HIP(ctx, s_minmax.alloc((size_t)chunks * 6 * sizeof(float)));
  float* d_minmax = s_minmax.as<float>();
  hipLaunchKernelGGL(chunk_bounds_kernel, dim3((unsigned)chunks), dim3(64), 0, ctx->stream, ps.d_xyzd,
                     (long long)ps.n, ps.d_chunk_bounds, d_minmax);
  // exact AABB of the point positions (feeds getSubmapFrameSurfaceObb for kVoxels points)
  std::vector<float> mm((size_t)chunks * 6);
  hipError_t e = hipGetLastError();
  if (e == hipSuccess) e = hipMemcpyAsync(mm.data(), d_minmax, mm.size() * sizeof(float), hipMemcpyDeviceToHost, ctx->stream);
  if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
  if (e != hipSuccess) return set_error(ctx, VGX_ERR_HIP, std::string("chunk bounds: ") + hipGetErrorString(e));
  for (int a = 0; a < 3; ++a) {
    ps.aabb_min[a] = mm[(size_t)a];
    ps.aabb_max[a] = mm[3 + (size_t)a];
  }
  for (long long c = 1; c < chunks; ++c)
    for (int a = 0; a < 3; ++a) {
      ps.aabb_min[a] = std::min(ps.aabb_min[a], mm[(size_t)c * 6 + a]);
      ps.aabb_max[a] = std::max(ps.aabb_max[a], mm[(size_t)c * 6 + 3 + a]);
    }
  return VGX_OK;
}

// Dense block lookup table over the AABB of sm->block_index (host copy), uploaded
// to sm->d_lut.  Stands in for voxblox's unordered_map<BlockIndex, Block::Ptr>.
int build_block_lut(vgx_submap sm) {
  vgx_ctx ctx = sm->ctx;
  const int n_blocks = sm->n_blocks;
  const int32_t* block_index = sm->block_index.data();
  int32_t mn[3] = {0, 0, 0}, mx[3] = {-1, -1, -1};
  for (int b = 0; b < n_blocks; ++b)
    for (int a = 0; a < 3; ++a) {
      int32_t v = block_index[3 * b + a];
      if (b == 0 || v < mn[a]) mn[a] = v;
      if (b == 0 || v > mx[a]) mx[a] = v;
    }
  size_t total = 1;
  for (int a = 0; a < 3; ++a) {
    sm->lut_min[a] = mn[a];
    sm->lut_dim[a] = mx[a] - mn[a] + 1;
    total *= (size_t)sm->lut_dim[a];
    if (total > ((size_t)1 << 28))
      return set_error(ctx, VGX_ERR_UNSUPPORTED,
                       "submap block AABB exceeds 2^28 cells (dense lookup table)");
  }
  if (n_blocks == 0) return VGX_OK;
  std::vector<int32_t> lut(total, -1);
  for (int b = 0; b < n_blocks; ++b) {
    size_t ix = (size_t)(block_index[3 * b + 0] - mn[0]);
    size_t iy = (size_t)(block_index[3 * b + 1] - mn[1]);
    size_t iz = (size_t)(block_index[3 * b + 2] - mn[2]);
    lut[ix + (size_t)sm->lut_dim[0] * (iy + (size_t)sm->lut_dim[1] * iz)] = b;
  }
  VGX_HIP(ctx, hipMalloc(&sm->d_lut, total * sizeof(int32_t)));
  VGX_HIP(ctx, hipMemcpy(sm->d_lut, lut.data(), total * sizeof(int32_t), hipMemcpyHostToDevice));
  return VGX_OK;
}

}  // namespace vgx

namespace vgx {
void reset_point_set(PointSet& ps) {
  static std::atomic<uint64_t> next_version{1};
  if (ps.d_xyzd) (void)hipFree(ps.d_xyzd);
  if (ps.d_weight) (void)hipFree(ps.d_weight);
  if (ps.d_chunk_bounds) (void)hipFree(ps.d_chunk_bounds);
  if (ps.d_cumulative) (void)hipFree(ps.d_cumulative);
  if (ps.d_search_lut) (void)hipFree(ps.d_search_lut);
  if (ps.d_inv_order) (void)hipFree(ps.d_inv_order);
  if (ps.rng.d_state) (void)hipFree(ps.rng.d_state);
  ps = PointSet();
  // cost functions and batches built on the old points notice the change (they hold raw device
  // pointers into the arrays freed above)
  ps.version = next_version.fetch_add(1);
}

int engine_to_host(vgx_ctx ctx, SamplerEngine& e) {
  if (!e.on_device) return VGX_OK;
  VGX_HIP(ctx, hipStreamSynchronize(ctx->stream));
  VGX_HIP(ctx, hipMemcpy(&e.host, e.d_state, sizeof(Mt19937), hipMemcpyDeviceToHost));
  e.on_device = false;
  return VGX_OK;
}

int engine_to_device(vgx_ctx ctx, SamplerEngine& e) {
  if (e.on_device) return VGX_OK;
  if (!e.d_state) VGX_HIP(ctx, hipMalloc(&e.d_state, sizeof(Mt19937)));
  // pageable source: the copy has left the host buffer when the call returns
  VGX_HIP(ctx, hipMemcpyAsync(e.d_state, &e.host, sizeof(Mt19937), hipMemcpyHostToDevice, ctx->stream));
  e.on_device = true;
  return VGX_OK;
}
}  // namespace vgx

using namespace vgx;

int vgx_submap_s::ensure_quad_grid(int which) {
  Grid& g = grid[which];
  if (!g.present || !g.d_bricks || g.layout != VGX_BRICKS_APRON || g.d_quad) return VGX_OK;
  const size_t cells = brick_cells(vps, VGX_BRICKS_QUAD);
  if ((unsigned long long)n_blocks * cells >= (1ull << 32))
    return set_error(ctx, VGX_ERR_UNSUPPORTED, "submap too large for 32-bit addressing of its quad bricks: " +
                                                   std::to_string(n_blocks) + " blocks");
  // (tests only: VGX_TEST_QUAD_ALLOC_FAILS_AFTER=k makes every allocation after the k-th fail, tests/test_brick_layout_gpu.py)
  static const long fail_after = getenv("VGX_TEST_QUAD_ALLOC_FAILS_AFTER") ? atol(getenv("VGX_TEST_QUAD_ALLOC_FAILS_AFTER")) : -1;
  static long made_so_far = 0;
  const bool pretend_oom = fail_after >= 0 && made_so_far++ >= fail_after;
  if (pretend_oom || hipMalloc(&g.d_quad, (size_t)n_blocks * cells * sizeof(float)) != hipSuccess) {
    (void)hipGetLastError();
    g.d_quad = nullptr;
    return set_error(ctx, VGX_ERR_NOMEM, "quad bricks for a sampling session: device allocation failed (" +
                                             std::to_string((size_t)n_blocks * cells * sizeof(float)) + " bytes; "
                                             "vgx_ctx_set_sampling_bricks(ctx, VGX_SAMPLING_BRICKS_SAME) does without)");
  }
  if (vps == 16)
    hipLaunchKernelGGL(requad_kernel<16>, dim3(n_blocks), dim3(256), 0, ctx->stream, g.d_bricks, g.d_quad);
  else
    hipLaunchKernelGGL(requad_kernel<8>, dim3(n_blocks), dim3(256), 0, ctx->stream, g.d_bricks, g.d_quad);
  VGX_HIP(ctx, hipGetLastError());
  return VGX_OK;
}

vgx::GridDev vgx_submap_s::grid_dev(int which) const {
  GridDev g;
  g.bricks = grid[which].d_bricks;
  g.layout = grid[which].layout;
  g.lut = d_lut;
  for (int a = 0; a < 3; ++a) {
    g.lut_min[a] = lut_min[a];
    g.lut_dim[a] = lut_dim[a];
  }
  g.voxel_size = voxel_size;
  g.voxel_size_inv = voxel_size_inv;
  g.block_size = block_size;
  g.block_size_inv = block_size_inv;
  return g;
}

// ---------------------------------------------------------------------------
// context
// ---------------------------------------------------------------------------
namespace vgx {
std::mutex& lifetime_mu() {
  static std::mutex mu;
  return mu;
}
}  // namespace vgx

extern "C" {

int vgx_ctx_create(int device, vgx_ctx* out) {
  if (!out) return set_error(nullptr, VGX_ERR_INVALID, "vgx_ctx_create: out == NULL");
  *out = nullptr;
  int count = 0;
  hipError_t e = hipGetDeviceCount(&count);
  if (e != hipSuccess || count <= 0)
    return set_error(nullptr, VGX_ERR_NO_DEVICE,
                     std::string("vgx_ctx_create: no HIP device (") +
                         (e != hipSuccess ? hipGetErrorString(e) : "count == 0") +
                         "); libvoxgraph_amd has no CPU fallback");
  if (device < 0 || device >= count)
    return set_error(nullptr, VGX_ERR_INVALID, "vgx_ctx_create: device index out of range");
  hipDeviceProp_t prop;
  if (hipGetDeviceProperties(&prop, device) != hipSuccess)
    return set_error(nullptr, VGX_ERR_HIP, "vgx_ctx_create: hipGetDeviceProperties failed");
  if (std::string(prop.gcnArchName).rfind("gfx950", 0) != 0)
    return set_error(nullptr, VGX_ERR_NO_DEVICE,
                     std::string("vgx_ctx_create: device is ") + prop.gcnArchName +
                         ", this library is built for gfx950 only");
  vgx_ctx ctx = new (std::nothrow) vgx_ctx_s();
  if (!ctx) return set_error(nullptr, VGX_ERR_NOMEM, "vgx_ctx_create: out of host memory");
  ctx->device = device;
  ctx->cu_count = prop.multiProcessorCount;
  // Who goes first when both sides have work: the TSDF side.  A scan is one short kernel (tens of microseconds) that the
  // sensor's cadence waits for; a solver evaluation is thousands of workgroups that nobody waits for individually
  // (voxgraph_mapper.cpp:218-238: optimisation runs in the background of the mapping thread).  So the TSDF stream is
  // created with the device's highest priority and the registration stream with its lowest: the scan's workgroups are
  // dispatched as soon as workgroups of the evaluation retire instead of queueing behind all of them
  // (profiles/r06_scan_latency.txt: per-scan latency under a running solve with and without).  VGX_STREAM_PRIORITY=0: both
  // at the default priority (A/B aid).
  int prio_least = 0, prio_greatest = 0;
  const char* prio_env = getenv("VGX_STREAM_PRIORITY");
  const bool use_prio = !(prio_env && atoi(prio_env) == 0) && hipSetDevice(device) == hipSuccess &&
                        hipDeviceGetStreamPriorityRange(&prio_least, &prio_greatest) == hipSuccess;
  ctx->stream_priorities = use_prio && prio_least != prio_greatest;
  if (hipSetDevice(device) != hipSuccess ||
      hipStreamCreateWithPriority(&ctx->own_stream, hipStreamNonBlocking, ctx->stream_priorities ? prio_least : 0) != hipSuccess ||
      hipStreamCreateWithPriority(&ctx->tsdf_own_stream, hipStreamNonBlocking, ctx->stream_priorities ? prio_greatest : 0) != hipSuccess ||
      hipEventCreate(&ctx->ev_start) != hipSuccess ||
      hipEventCreate(&ctx->ev_stop) != hipSuccess ||
      hipEventCreate(&ctx->ev_tsdf_start) != hipSuccess ||
      hipEventCreate(&ctx->ev_tsdf_stop) != hipSuccess ||
      hipEventCreateWithFlags(&ctx->ev_handover, hipEventDisableTiming) != hipSuccess) {
    if (ctx->own_stream) (void)hipStreamDestroy(ctx->own_stream);
    if (ctx->tsdf_own_stream) (void)hipStreamDestroy(ctx->tsdf_own_stream);
    for (hipEvent_t ev : {ctx->ev_start, ctx->ev_stop, ctx->ev_tsdf_start, ctx->ev_tsdf_stop, ctx->ev_handover})
      if (ev) (void)hipEventDestroy(ev);
    delete ctx;
    return set_error(nullptr, VGX_ERR_HIP, "vgx_ctx_create: stream/event creation failed");
  }
  ctx->stream = ctx->own_stream;
  ctx->tsdf_stream = ctx->tsdf_own_stream;
  ctx->last_error = "no error";
  *out = ctx;
  return VGX_OK;
}

int vgx_ctx_destroy(vgx_ctx ctx) {
  if (!ctx) return VGX_ERR_INVALID;
  (void)hipSetDevice(ctx->device);
  (void)hipStreamSynchronize(ctx->stream);
  (void)hipStreamSynchronize(ctx->tsdf_stream);
  for (hipEvent_t ev : {ctx->ev_start, ctx->ev_stop, ctx->ev_tsdf_start, ctx->ev_tsdf_stop, ctx->ev_handover})
    if (ev) (void)hipEventDestroy(ev);
  for (int k = 0; k < Context::kEvalSlots; ++k) {
    Context::EvalSlot& sl = ctx->eval_slot[k];
    if (sl.stream) {
      (void)hipStreamSynchronize(sl.stream);
      (void)hipStreamDestroy(sl.stream);
    }
    if (sl.order) (void)hipEventDestroy(sl.order);
    if (sl.d_out) (void)hipFree(sl.d_out);
    if (sl.d_raw) (void)hipFree(sl.d_raw);
    if (sl.h_raw) (void)hipHostFree(sl.h_raw);
    if (sl.h_out) (void)hipHostFree(sl.h_out);
  }
  if (ctx->own_stream) (void)hipStreamDestroy(ctx->own_stream);
  if (ctx->tsdf_own_stream) (void)hipStreamDestroy(ctx->tsdf_own_stream);
  delete ctx;
  return VGX_OK;
}

const char* vgx_last_error(vgx_ctx ctx) {
  // a per-thread copy: the returned pointer stays valid while other threads record new errors
  static thread_local std::string copy;
  if (ctx) {
    std::lock_guard<std::mutex> lk(ctx->err_mu);
    copy = ctx->last_error;
  } else {
    std::lock_guard<std::mutex> lk(g_err_mu);
    copy = g_global_error;
  }
  return copy.c_str();
}

int vgx_ctx_set_stream(vgx_ctx ctx, void* hip_stream) {
  if (!ctx) return VGX_ERR_INVALID;
  std::lock_guard<std::mutex> lk(ctx->mu);
  ctx->stream = hip_stream ? (hipStream_t)hip_stream : ctx->own_stream;
  return VGX_OK;
}

void* vgx_ctx_get_stream(vgx_ctx ctx) { return ctx ? (void*)ctx->stream : nullptr; }

int vgx_ctx_set_tsdf_stream(vgx_ctx ctx, void* hip_stream) {
  if (!ctx) return VGX_ERR_INVALID;
  std::lock_guard<std::mutex> lk(ctx->tsdf_mu);
  (void)hipSetDevice(ctx->device);
  (void)hipStreamSynchronize(ctx->tsdf_stream);  // what was queued on the old stream is done before the new one is used
  ctx->tsdf_stream = hip_stream ? (hipStream_t)hip_stream : ctx->tsdf_own_stream;
  return VGX_OK;
}

void* vgx_ctx_get_tsdf_stream(vgx_ctx ctx) { return ctx ? (void*)ctx->tsdf_stream : nullptr; }

int vgx_ctx_set_brick_layout(vgx_ctx ctx, int32_t layout) {
  if (!ctx) return VGX_ERR_INVALID;
  if (layout != VGX_BRICKS_APRON && layout != VGX_BRICKS_QUAD)
    return set_error(ctx, VGX_ERR_INVALID, "vgx_ctx_set_brick_layout: unknown layout");
  std::lock_guard<std::mutex> lk(ctx->mu);
  ctx->brick_layout = layout;
  return VGX_OK;
}

int vgx_ctx_set_sampling_bricks(vgx_ctx ctx, int32_t mode) {
  if (!ctx) return VGX_ERR_INVALID;
  if (mode != VGX_SAMPLING_BRICKS_SAME && mode != VGX_SAMPLING_BRICKS_QUAD)
    return set_error(ctx, VGX_ERR_INVALID, "vgx_ctx_set_sampling_bricks: unknown mode");
  std::lock_guard<std::mutex> lk(ctx->mu);
  ctx->sampling_bricks = mode;
  return VGX_OK;
}

int vgx_ctx_synchronize(vgx_ctx ctx) {
  if (!ctx) return VGX_ERR_INVALID;
  VGX_HIP(ctx, hipStreamSynchronize(ctx->stream));
  VGX_HIP(ctx, hipStreamSynchronize(ctx->tsdf_stream));
  return VGX_OK;
}

// the TSDF side alone: the mapping thread's "is my scan in?" without waiting for a solver evaluation on the other stream
int vgx_ctx_synchronize_tsdf(vgx_ctx ctx) {
  if (!ctx) return VGX_ERR_INVALID;
  hipStream_t st;
  {
    std::lock_guard<std::mutex> lk(ctx->tsdf_mu);
    st = ctx->tsdf_stream;
  }
  VGX_HIP(ctx, hipStreamSynchronize(st));
  return VGX_OK;
}

// The TSDF stream waits, on the device, for what `producer_stream` holds now (NULL: the context's registration stream): a
// caller whose scan points are PRODUCED on another stream (a driver, PyTorch, the registration side) orders its
// vgx_tsdf_integrate*_device calls behind that producer without a host synchronisation.
int vgx_ctx_tsdf_wait_for_stream(vgx_ctx ctx, void* producer_stream) {
  if (!ctx) return VGX_ERR_INVALID;
  hipStream_t producer = (hipStream_t)producer_stream;
  if (!producer) {
    std::lock_guard<std::mutex> lk(ctx->mu);
    producer = ctx->stream;
  }
  std::lock_guard<std::mutex> lk(ctx->tsdf_mu);
  VGX_HIP(ctx, hipSetDevice(ctx->device));
  hipEvent_t ev = nullptr;
  VGX_HIP(ctx, hipEventCreateWithFlags(&ev, hipEventDisableTiming));
  hipError_t e = hipEventRecord(ev, producer);
  if (e == hipSuccess) e = hipStreamWaitEvent(ctx->tsdf_stream, ev, 0);
  (void)hipEventDestroy(ev);   // (destroyed once the wait has consumed it: HIP defers the release)
  VGX_HIP(ctx, e);
  return VGX_OK;
}

int vgx_ctx_stream_priorities(vgx_ctx ctx) { return ctx && ctx->stream_priorities ? 1 : 0; }

// The timer brackets BOTH streams of the context (registration side and TSDF side): elapsed = the longer of the two
// start-to-stop intervals, i.e. the time until everything enqueued in between is done, whichever stream it went to.
int vgx_ctx_timer_start(vgx_ctx ctx) {
  if (!ctx) return VGX_ERR_INVALID;
  VGX_HIP(ctx, hipEventRecord(ctx->ev_start, ctx->stream));
  {
    std::lock_guard<std::mutex> lk(ctx->tsdf_mu);
    VGX_HIP(ctx, hipEventRecord(ctx->ev_tsdf_start, ctx->tsdf_stream));
  }
  return VGX_OK;
}

int vgx_ctx_timer_stop(vgx_ctx ctx, float* elapsed_ms) {
  if (!ctx || !elapsed_ms) return VGX_ERR_INVALID;
  VGX_HIP(ctx, hipEventRecord(ctx->ev_stop, ctx->stream));
  {
    std::lock_guard<std::mutex> lk(ctx->tsdf_mu);
    VGX_HIP(ctx, hipEventRecord(ctx->ev_tsdf_stop, ctx->tsdf_stream));
  }
  VGX_HIP(ctx, hipEventSynchronize(ctx->ev_stop));
  VGX_HIP(ctx, hipEventSynchronize(ctx->ev_tsdf_stop));
  float a = 0.0f, b = 0.0f;
  VGX_HIP(ctx, hipEventElapsedTime(&a, ctx->ev_start, ctx->ev_stop));
  VGX_HIP(ctx, hipEventElapsedTime(&b, ctx->ev_tsdf_start, ctx->ev_tsdf_stop));
  *elapsed_ms = a > b ? a : b;
  return VGX_OK;
}

// ---------------------------------------------------------------------------
// submaps
// ---------------------------------------------------------------------------
static int upload(vgx_ctx ctx, const void* src, size_t bytes, void** dst) {
  *dst = nullptr;
  if (!src || bytes == 0) return VGX_OK;
  VGX_HIP(ctx, hipMalloc(dst, bytes));
  VGX_HIP(ctx, hipMemcpyAsync(*dst, src, bytes, hipMemcpyHostToDevice, ctx->stream));
  return VGX_OK;
}

int vgx_submap_create(vgx_ctx ctx, int32_t submap_id, float voxel_size, int32_t vps,
                      int32_t n_blocks, const int32_t* block_index, const float* tsdf_distance,
                      const float* tsdf_weight, const float* esdf_distance,
                      const uint8_t* esdf_observed, vgx_submap* out) {
  if (!ctx || !out) return VGX_ERR_INVALID;
  *out = nullptr;
  std::lock_guard<std::mutex> lk(ctx->mu);
  if (!(voxel_size > 0) || n_blocks < 0 || (n_blocks > 0 && !block_index))
    return set_error(ctx, VGX_ERR_INVALID, "vgx_submap_create: bad voxel_size / n_blocks / block_index");
  if (vps != 16 && vps != 8)
    return set_error(ctx, VGX_ERR_UNSUPPORTED, "vgx_submap_create: voxels_per_side must be 8 or 16");
  if ((tsdf_distance == nullptr) != (tsdf_weight == nullptr) ||
      (esdf_observed != nullptr && esdf_distance == nullptr))
    return set_error(ctx, VGX_ERR_INVALID,
                     "vgx_submap_create: tsdf_distance/tsdf_weight come together; esdf_observed needs esdf_distance");
  VGX_HIP(ctx, hipSetDevice(ctx->device));
  vgx_submap sm = new (std::nothrow) vgx_submap_s();
  if (!sm) return set_error(ctx, VGX_ERR_NOMEM, "vgx_submap_create: out of host memory");
  sm->ctx = ctx;
  sm->id = submap_id;
  sm->vps = vps;
  sm->n_blocks = n_blocks;
  // voxblox::Layer / Block constants, f32 [recalled]
  sm->voxel_size = voxel_size;
  sm->voxel_size_inv = 1.0f / voxel_size;
  sm->block_size = (float)vps * voxel_size;
  sm->block_size_inv = 1.0f / sm->block_size;
  sm->block_index.assign(block_index, block_index + 3 * (size_t)n_blocks);

  int rc = build_block_lut(sm);
  if (rc == VGX_OK && n_blocks > 0) {
    const size_t nvox = (size_t)n_blocks * vps * vps * vps;
    rc = upload(ctx, block_index, 3 * (size_t)n_blocks * sizeof(int32_t), (void**)&sm->d_block_index);
    if (rc == VGX_OK) rc = upload(ctx, tsdf_distance, nvox * sizeof(float), (void**)&sm->d_tsdf_distance);
    if (rc == VGX_OK) rc = upload(ctx, tsdf_weight, nvox * sizeof(float), (void**)&sm->d_tsdf_weight);
    if (rc == VGX_OK) rc = upload(ctx, esdf_distance, nvox * sizeof(float), (void**)&sm->d_esdf_distance);
    if (rc == VGX_OK) rc = upload(ctx, esdf_observed, nvox * sizeof(uint8_t), (void**)&sm->d_esdf_observed);
    // the H2D copies above read pageable host memory: finish them before the
    // caller's arrays go away
    if (rc == VGX_OK && hipStreamSynchronize(ctx->stream) != hipSuccess)
      rc = set_error(ctx, VGX_ERR_HIP, "vgx_submap_create: H2D upload failed");
    if (rc == VGX_OK && sm->d_tsdf_distance) rc = launch_brickify(sm, 0);
    if (rc == VGX_OK && sm->d_esdf_distance && sm->d_esdf_observed) rc = launch_brickify(sm, 1);
  }
  if (rc != VGX_OK) {
    vgx_submap_destroy(sm);
    return rc;
  }
  *out = sm;
  return VGX_OK;
}


int vgx_submap_release_raw_layers(vgx_submap sm) {
  if (!sm) return VGX_ERR_INVALID;
  (void)hipStreamSynchronize(sm->ctx->stream);
  if (sm->d_tsdf_distance) (void)hipFree(sm->d_tsdf_distance);
  if (sm->d_tsdf_weight) (void)hipFree(sm->d_tsdf_weight);
  if (sm->d_esdf_distance) (void)hipFree(sm->d_esdf_distance);
  if (sm->d_esdf_observed) (void)hipFree(sm->d_esdf_observed);
  sm->d_tsdf_distance = sm->d_tsdf_weight = sm->d_esdf_distance = nullptr;
  sm->d_esdf_observed = nullptr;
  return VGX_OK;
}

int vgx_submap_download_layers(vgx_submap sm, float* tsdf_distance, float* tsdf_weight,
                               float* esdf_distance, uint8_t* esdf_observed) {
  if (!sm) return VGX_ERR_INVALID;
  vgx_ctx ctx = sm->ctx;
  VGX_HIP(ctx, hipSetDevice(ctx->device));
  VGX_HIP(ctx, hipStreamSynchronize(ctx->stream));
  const size_t nvox = (size_t)sm->n_blocks * sm->vps * sm->vps * sm->vps;
  struct { void* dst; const void* src; size_t bytes; } jobs[4] = {
      {tsdf_distance, sm->d_tsdf_distance, nvox * sizeof(float)},
      {tsdf_weight, sm->d_tsdf_weight, nvox * sizeof(float)},
      {esdf_distance, sm->d_esdf_distance, nvox * sizeof(float)},
      {esdf_observed, sm->d_esdf_observed, nvox * sizeof(uint8_t)}};
  for (auto& j : jobs) {
    if (!j.dst) continue;
    if (!j.src) return set_error(ctx, VGX_ERR_INVALID, "vgx_submap_download_layers: layer not resident");
    if (j.bytes) VGX_HIP(ctx, hipMemcpy(j.dst, j.src, j.bytes, hipMemcpyDeviceToHost));
  }
  return VGX_OK;
}

int vgx_submap_block_index(vgx_submap sm, int32_t* block_index) {
  if (!sm || !block_index) return VGX_ERR_INVALID;
  std::copy(sm->block_index.begin(), sm->block_index.end(), block_index);
  return VGX_OK;
}

int vgx_submap_destroy(vgx_submap sm) {
  if (!sm) return VGX_ERR_INVALID;
  {
    // cost functions still read this submap's bricks and points: the last of them to go frees it
    std::lock_guard<std::mutex> lk(vgx::lifetime_mu());
    if (sm->users > 0) {
      sm->destroy_requested = true;
      return VGX_OK;
    }
  }
  (void)hipSetDevice(sm->ctx->device);
  vgx_submap_release_raw_layers(sm);
  if (sm->d_lut) (void)hipFree(sm->d_lut);
  if (sm->d_block_index) (void)hipFree(sm->d_block_index);
  if (sm->d_iso_block_index) (void)hipFree(sm->d_iso_block_index);
  for (int k = 0; k < 2; ++k) {
    if (sm->grid[k].d_bricks) (void)hipFree(sm->grid[k].d_bricks);
    if (sm->grid[k].d_quad) (void)hipFree(sm->grid[k].d_quad);
    reset_point_set(sm->points[k]);
  }
  delete sm;
  return VGX_OK;
}

int32_t vgx_submap_id(vgx_submap sm) { return sm ? sm->id : -1; }
int32_t vgx_submap_num_blocks(vgx_submap sm) { return sm ? sm->n_blocks : -1; }

// Morton code of non-negative 21-bit voxel coordinates
static inline uint64_t spread3(uint64_t v) {
  v &= 0x1fffff;
  v = (v | v << 32) & 0x1f00000000ffffULL;
  v = (v | v << 16) & 0x1f0000ff0000ffULL;
  v = (v | v << 8) & 0x100f00f00f00f00fULL;
  v = (v | v << 4) & 0x10c30c30c30c30c3ULL;
  v = (v | v << 2) & 0x1249249249249249ULL;
  return v;
}

int vgx_submap_set_points(vgx_submap sm, int32_t point_type, int64_t n, const float* xyz,
                          const float* distance, const float* weight, uint32_t flags) {
  if (!sm) return VGX_ERR_INVALID;
  vgx_ctx ctx = sm->ctx;
  std::lock_guard<std::mutex> lk(ctx->mu);
  if (point_type != VGX_POINTS_ISOSURFACE && point_type != VGX_POINTS_VOXELS)
    return set_error(ctx, VGX_ERR_INVALID, "vgx_submap_set_points: bad point_type");
  if (n < 0 || n > INT32_MAX || (n > 0 && (!xyz || !distance || !weight)))
    return set_error(ctx, VGX_ERR_INVALID, "vgx_submap_set_points: bad n or NULL arrays");
  VGX_HIP(ctx, hipSetDevice(ctx->device));
  PointSet& ps = sm->points[point_type];
  (void)hipStreamSynchronize(ctx->stream);
  reset_point_set(ps);
  ps.n = n;
  // WeightedSampler::addItem (weighted_sampler_inl.h:5-16): cumulative weights
  // in upload order, accumulated in double
  ps.cumulative_weight.resize((size_t)n);
  double acc = 0;
  for (int64_t i = 0; i < n; ++i) {
    acc = (i == 0) ? (double)weight[i] : acc + (double)weight[i];
    ps.cumulative_weight[(size_t)i] = acc;
  }
  // summed_reference_weight of the deterministic mode (.cpp:124): same order of
  // f64 additions as the reference's loop when the order is kept; the Morton
  // order changes the rounding by O(n eps) only.
  std::vector<int64_t> order;
  if ((flags & VGX_POINTS_SORT_MORTON) && n > 1) {
    std::vector<uint64_t> key((size_t)n);
    float mnv[3] = {xyz[0], xyz[1], xyz[2]};
    for (int64_t i = 0; i < n; ++i)
      for (int a = 0; a < 3; ++a) mnv[a] = std::min(mnv[a], xyz[3 * i + a]);
    // align the curve to the voxel grid so that 2^k-voxel cells nest in blocks
    float base[3];
    for (int a = 0; a < 3; ++a) base[a] = std::floor(mnv[a] * sm->block_size_inv) * sm->block_size;
    for (int64_t i = 0; i < n; ++i) {
      uint64_t c[3];
      for (int a = 0; a < 3; ++a) {
        double v = std::floor(((double)xyz[3 * i + a] - (double)base[a]) * (double)sm->voxel_size_inv);
        c[a] = (uint64_t)std::min(std::max(v, 0.0), 2097151.0);
      }
      key[(size_t)i] = spread3(c[0]) | (spread3(c[1]) << 1) | (spread3(c[2]) << 2);
    }
    order.resize((size_t)n);
    std::iota(order.begin(), order.end(), (int64_t)0);
    std::stable_sort(order.begin(), order.end(),
                     [&](int64_t a, int64_t b) { return key[(size_t)a] < key[(size_t)b]; });
  }
  std::vector<float4> h_xyzd((size_t)n);
  std::vector<float> h_w((size_t)n);
  double sum_w = 0;
  for (int64_t i = 0; i < n; ++i) {
    int64_t s = order.empty() ? i : order[(size_t)i];
    h_xyzd[(size_t)i] = make_float4(xyz[3 * s], xyz[3 * s + 1], xyz[3 * s + 2], distance[s]);
    h_w[(size_t)i] = weight[s];
    sum_w += (double)weight[s];
  }
  ps.sum_weight = sum_w;
  ps.order = order;
  if (!order.empty()) {
    ps.inv_order.resize((size_t)n);
    for (int64_t i = 0; i < n; ++i) ps.inv_order[(size_t)order[(size_t)i]] = (int32_t)i;
  }
  if (n > 0) {
    VGX_HIP(ctx, hipMalloc(&ps.d_xyzd, (size_t)n * sizeof(float4)));
    VGX_HIP(ctx, hipMalloc(&ps.d_weight, (size_t)n * sizeof(float)));
    VGX_HIP(ctx, hipMemcpy(ps.d_xyzd, h_xyzd.data(), (size_t)n * sizeof(float4), hipMemcpyHostToDevice));
    VGX_HIP(ctx, hipMemcpy(ps.d_weight, h_w.data(), (size_t)n * sizeof(float), hipMemcpyHostToDevice));
  }
  const int rc = build_chunk_bounds(ctx, ps);
  ps.present = rc == VGX_OK;  // only a completely built set is offered to cost functions
  return rc;
}

int64_t vgx_submap_num_points(vgx_submap sm, int32_t point_type) {
  if (!sm || point_type < 0 || point_type > 1 || !sm->points[point_type].present) return -1;
  return sm->points[point_type].n;
}

int vgx_submap_point_order(vgx_submap sm, int32_t point_type, int64_t* order) {
  if (!sm || point_type < 0 || point_type > 1 || !order) return VGX_ERR_INVALID;
  const PointSet& ps = sm->points[point_type];
  if (!ps.present) return set_error(sm->ctx, VGX_ERR_INVALID, "vgx_submap_point_order: no such point set");
  for (int64_t i = 0; i < ps.n; ++i) order[i] = ps.order.empty() ? i : ps.order[(size_t)i];
  return VGX_OK;
}

int vgx_submap_download_points(vgx_submap sm, int32_t point_type, float* xyz, float* distance,
                               float* weight) {
  if (!sm || point_type < 0 || point_type > 1) return VGX_ERR_INVALID;
  vgx_ctx ctx = sm->ctx;
  const PointSet& ps = sm->points[point_type];
  if (!ps.present) return set_error(ctx, VGX_ERR_INVALID, "vgx_submap_download_points: no such point set");
  if (ps.n == 0) return VGX_OK;
  VGX_HIP(ctx, hipSetDevice(ctx->device));
  VGX_HIP(ctx, hipStreamSynchronize(ctx->stream));
  std::vector<float4> h((size_t)ps.n);
  VGX_HIP(ctx, hipMemcpy(h.data(), ps.d_xyzd, (size_t)ps.n * sizeof(float4), hipMemcpyDeviceToHost));
  for (int64_t i = 0; i < ps.n; ++i) {
    if (xyz) {
      xyz[3 * i] = h[(size_t)i].x;
      xyz[3 * i + 1] = h[(size_t)i].y;
      xyz[3 * i + 2] = h[(size_t)i].z;
    }
    if (distance) distance[i] = h[(size_t)i].w;
  }
  if (weight) VGX_HIP(ctx, hipMemcpy(weight, ps.d_weight, (size_t)ps.n * sizeof(float), hipMemcpyDeviceToHost));
  return VGX_OK;
}

}  // extern "C"

// the three hooks of the tooling library (vgx_internal.h)
extern "C" {
int vgx_internal_set_error(vgx_ctx ctx, int code, const char* msg) { return vgx::set_error(ctx, code, msg ? msg : ""); }
int vgx_internal_launch_brickify(vgx_submap sm, int which) { return vgx::launch_brickify(sm, which); }
int vgx_internal_build_block_lut(vgx_submap sm) { return vgx::build_block_lut(sm); }
}
