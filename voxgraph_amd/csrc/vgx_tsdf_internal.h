// Shared pieces of the TSDF path (vgx_tsdf.hip: the racing integrators and the layer;
// vgx_tsdf_det.hip: the reproducible integration mode).  gfx950 only.
#ifndef VGX_TSDF_INTERNAL_H_
#define VGX_TSDF_INTERNAL_H_

#include <cmath>
#include <cstdlib>
#include <cstring>
#include <utility>
#include <vector>

#include <rocprim/rocprim.hpp>

#include "vgx_internal.h"

namespace vgx {

struct TsdfLayerDev {
  unsigned long long* voxels;  // [max_blocks][vps^3] {distance (lo), weight (hi)}
  uint32_t* rgba;              // [max_blocks][vps^3]
  int32_t* lut;                // dense [dim.z][dim.y][dim.x]: slot, -1 free, -2 being allocated, -3 pool exhausted
  int32_t* block_index;        // [max_blocks][3]
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
  unsigned long long* n_updates;  // [kScanStatWords]: voxel updates, then the walk statistics of a counted scan
  unsigned long long* wg_stats;   // counted scans of the cooperative kernel: [workgroups][kWgStatWords]
  // the event log of a traced racing scan (vgx_tsdf_coop_kernel.h; null in every scan of the product library: the
  // logging instantiations of the kernel live in libvoxgraph_amd_bench.so, vgx_tsdf_integrator_set_event_trace)
  unsigned long long* trace;
  unsigned long long trace_words;
};
// a workgroup's row: 0..3 wall_clock64 at its start / rays queued / walk done / end, 4 rays, 5 rounds, 6 per-voxel folds,
// 7 longest chain of repeated folds, 8..15 what reduce_wg_stats_kernel sums into n_updates[0..7]
constexpr int kWgStatWords = 16;
constexpr int kScanStatWords = 8;

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

// Color::blendTwoColors [recalled]
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

// RayCaster [recalled, voxblox integrator_utils.h] set up from sensor origin t and end point g (both
// layer frame): ray_start / ray_end as the constructor computes them, then setupRayCaster on the
// scaled ends -- from the far end towards the origin (cast_from_origin = false, the fast
// integrator) or from the origin outwards (true, the merged and simple integrators).
// getGridIndexFromPoint's cast (oracle/tsdf_oracle.c grid_index): floor to int, NaN -> 0, saturating.  The reference's
// cast is undefined for such input; here both sides define it the same way, so a driver's "no return" code (inf, 1e30)
// gives the same start-set slot / group key in the oracle and on the device.
__device__ __forceinline__ int grid_index(float x) {
  x = floorf(x);
  if (!(x == x)) return 0;
  if (x >= 2147483648.0f) return 2147483647;
  if (x < -2147483648.0f) return -2147483647 - 1;
  return (int)x;
}

struct RayDda {
  int curr[3], sign[3];
  float t_next[3], t_step[3];
  long long steps;
  bool bad;
};

__device__ __forceinline__ RayDda ray_setup(const vgx_tsdf_config& c, float vsi, float tx, float ty, float tz,
                                            float gx, float gy, float gz, bool is_clearing,
                                            bool cast_from_origin) {
  const float dx = gx - tx, dy = gy - ty, dz = gz - tz;
  const float len = norm3(dx, dy, dz);
  const float ux = dx / len, uy = dy / len, uz = dz / len;
  const float trunc = c.default_truncation_distance;
  float sxx, syy, szz, exx, eyy, ezz;  // ray_start, ray_end
  if (is_clearing) {
    const float ray_length = fminf(fmaxf(len - trunc, 0.0f), c.max_ray_length_m);
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
  const float a_[3] = {sxx * vsi, syy * vsi, szz * vsi};  // start_scaled
  const float b_[3] = {exx * vsi, eyy * vsi, ezz * vsi};  // end_scaled
  RayDda r;
  r.bad = false;
  r.steps = 0;
#pragma unroll
  for (int a = 0; a < 3; ++a) {
    const float ss = cast_from_origin ? a_[a] : b_[a];
    const float es = cast_from_origin ? b_[a] : a_[a];
    r.bad |= (ss != ss) | (es != es);
    r.curr[a] = grid_index(ss + 1e-6f);
    const int end_index = grid_index(es + 1e-6f);
    const int diff = end_index - r.curr[a];
    r.steps += diff < 0 ? -diff : diff;
    const float ray_scaled = es - ss;
    r.sign[a] = signum(ray_scaled);
    const float corrected = (float)(r.sign[a] > 0 ? r.sign[a] : 0);
    const float dist_b = corrected - (ss - (float)r.curr[a]);
    if (ray_scaled == 0.0f) {
      r.t_next[a] = INFINITY;
      r.t_step[a] = INFINITY;
    } else {
      r.t_next[a] = dist_b / ray_scaled;
      r.t_step[a] = (float)r.sign[a] / ray_scaled;
    }
  }
  return r;
}

// RayCaster::nextRayIndex: advance along the axis with the smallest t (minCoeff: first minimum)
__device__ __forceinline__ void dda_advance(RayDda& r) {
  int m = 0;
  if (r.t_next[1] < r.t_next[m]) m = 1;
  if (r.t_next[2] < r.t_next[m]) m = 2;
  r.curr[0] += m == 0 ? r.sign[0] : 0; r.curr[1] += m == 1 ? r.sign[1] : 0; r.curr[2] += m == 2 ? r.sign[2] : 0;
  r.t_next[0] += m == 0 ? r.t_step[0] : 0.0f; r.t_next[1] += m == 1 ? r.t_step[1] : 0.0f;
  r.t_next[2] += m == 2 ? r.t_step[2] : 0.0f;
}

// MixedThreadSafeIndex [recalled]: 1024-point groups visited round-robin, the tail in order
__device__ __forceinline__ long long mixed_order_point(long long seq, long long n) {
  const long long step_size = 1024, number_of_groups = n / step_size;
  if (seq < number_of_groups * step_size) return (seq % number_of_groups) * step_size + seq / number_of_groups;
  return seq;
}
// ThreadSafeIndex::getNextIndex: the point visited seq-th -- SortedThreadSafeIndex's table when there is one
// (vgx_tsdf_config.integration_order = VGX_TSDF_ORDER_SORTED), else MixedThreadSafeIndex
__device__ __forceinline__ long long visiting_order_point(const uint32_t* __restrict__ order, long long seq, long long n) {
  return order ? (long long)order[seq] : mixed_order_point(seq, n);
}

// kindr::minimal transform of a sensor-frame point: Eigen _transformVector + translation
__device__ __forceinline__ void transform_point(float qw, float qx, float qy, float qz, float tx, float ty, float tz,
                                                float px, float py, float pz, float& gx, float& gy, float& gz) {
  float uvx = qy * pz - qz * py, uvy = qz * px - qx * pz, uvz = qx * py - qy * px;
  uvx += uvx; uvy += uvy; uvz += uvz;
  const float ccx = qy * uvz - qz * uvy, ccy = qz * uvx - qx * uvz, ccz = qx * uvy - qy * uvx;
  gx = (px + qw * uvx + ccx) + tx;
  gy = (py + qw * uvy + ccy) + ty;
  gz = (pz + qw * uvz + ccz) + tz;
}

// ---- prefix sums inside ONE launch: workgroups chained by epoch-tagged words ------------------------------------
// A scan at these sizes (10^4 .. 10^6 items) is a launch's latency, and rocprim's is two launches; the kernels that
// need a prefix (ray offsets, group ranks, compaction offsets) take it from here and go straight on with what
// the prefix was for.  Tiles are taken in the order the workgroups start (a ticket: a tile only waits for tiles
// that are already running), every tile publishes one 8-byte word
//     epoch << 34 | status << 32 | value        status 1: the tile's own sum, 2: the sum up to the tile's end
// and wave 0 looks back over the 64 tiles before it at a time until it meets a status-2 word.  The epoch is the
// launch's number (one counter per scratch object), so no word is ever cleared; the ticket counts on from `base`,
// what the host has launched on that counter since the scan's first kernel zeroed it.  (CDNA4: the words are the
// whole exchange between workgroups -- relaxed agent-scope 8-byte loads / stores, tag and value in one word.)
struct TileChain {
  unsigned long long* state;   // [tiles]
  unsigned long long epoch;    // 1 .. 2^30 - 1
  unsigned long long* ticket;  // counts workgroups as they start
  uint32_t ticket_base;
  unsigned long long* error;   // set to 3 if a tile that started before this one never reported (internal error)
};
constexpr unsigned long long kChainEpochMax = (1ull << 30) - 1;

// this workgroup's tile; every thread calls it (sh: one shared word)
__device__ __forceinline__ uint32_t chain_tile(const TileChain& ch, uint32_t* sh) {
  if (threadIdx.x == 0) *sh = (uint32_t)atomicAdd(ch.ticket, 1ull) - ch.ticket_base;
  __syncthreads();
  const uint32_t tile = *sh;
  __syncthreads();  // (sh is free again)
  return tile;
}

// the sum of `aggregate` over the tiles before `tile`; every thread of a 256-thread workgroup calls it with the
// same arguments (sh: one shared word)
__device__ __forceinline__ uint32_t chain_exclusive_sum(const TileChain& ch, uint32_t tile, uint32_t aggregate, uint32_t* sh) {
  if (threadIdx.x < 64) {
    const int lane = (int)threadIdx.x;
    const unsigned long long tag = ch.epoch << 34;
    if (lane == 0)
      __hip_atomic_store(ch.state + tile, tag | ((tile == 0 ? 2ull : 1ull) << 32) | aggregate, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    uint32_t prefix = 0;
    for (long long hi = (long long)tile - 1; hi >= 0; hi -= 64) {
      const long long t = hi - lane;
      unsigned long long x = 0;
      if (t >= 0) {
        unsigned spins = 0;
        while (((x = __hip_atomic_load(ch.state + t, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) >> 34) != ch.epoch) {
          __builtin_amdgcn_s_sleep(1);
          if (++spins > (1u << 22)) {  // (seconds)
            *ch.error = 3ull;
            x = tag | (2ull << 32);
            break;
          }
        }
      }
      const bool inclusive = t >= 0 && ((x >> 32) & 3ull) == 2ull;
      const unsigned long long m = __ballot(inclusive);
      uint32_t c = t >= 0 ? (uint32_t)x : 0u;
      if (m != 0ull && lane > __ffsll((long long)m) - 1) c = 0u;  // beyond the nearest inclusive word: already in it
#pragma unroll
      for (int d = 32; d > 0; d >>= 1) c += (uint32_t)__shfl_xor((int)c, d);
      prefix += c;
      if (m != 0ull) break;
    }
    if (lane == 0) {
      if (tile != 0)
        __hip_atomic_store(ch.state + tile, tag | (2ull << 32) | (unsigned long long)(uint32_t)(prefix + aggregate), __ATOMIC_RELAXED,
                           __HIP_MEMORY_SCOPE_AGENT);
      *sh = prefix;
    }
  }
  __syncthreads();
  const uint32_t prefix = *sh;
  __syncthreads();
  return prefix;
}

// exclusive prefix of v over the 256 threads of the workgroup, and their total (sh4: four shared words)
__device__ __forceinline__ uint32_t block_exclusive_sum(uint32_t v, uint32_t* sh4, uint32_t& total) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  uint32_t inc = v;
#pragma unroll
  for (int d = 1; d < 64; d <<= 1) {
    const uint32_t o = (uint32_t)__shfl_up((int)inc, d);
    if (lane >= d) inc += o;
  }
  if (lane == 63) sh4[wave] = inc;
  __syncthreads();
  uint32_t before = inc - v;
  total = 0;
#pragma unroll
  for (int w = 0; w < 4; ++w) {
    if (w < wave) before += sh4[w];
    total += sh4[w];
  }
  __syncthreads();
  return before;
}

// ---- the reproducible mode's / merged integrator's per-scan counters (u64 each; DetScratch::d_ctr) ----
// kCtrArrive: the ray kernel's blocks as they finish (low word) and how many of them moved a stopping step
// (high word); kCtrTicket: the sweep kernel's tiles in the order they start.  Both are back at zero when a
// sweep's ray kernel ends (its last block) and at the start of every scan (det_points_kernel).
// kCtrTotal: the accesses det_count_kernel counted (64 bits: an overflowing scan is seen, not wrapped); kCtrM /
// kCtrBlocks: the updates and the layer's block count as det_blocks_kernel found them (so that one copy of the
// counters is the whole read-back of a commit).  kCtrChainTicket: TileChain's ticket for the scan's prefix-sum
// kernels (never reset inside a scan: the host passes what it has launched so far as the base).  kCtrGroups /
// kCtrGroupPoints: the merged integrator's groups and the valid points in them (how many lanes the NEXT scan's
// merge kernel gives a group: vgx_tsdf_integrator_s::merged_lanes).
enum { kCtrChanged = 0, kCtrNew = 1, kCtrError = 2, kCtrDropped = 3, kCtrLong = 4, kCtrOverflow = 5, kCtrArrive = 6, kCtrTicket = 7,
       kCtrTotal = 8, kCtrM = 9, kCtrBlocks = 10, kCtrChainTicket = 11, kCtrGroups = 12, kCtrGroupPoints = 13,
       kCtrCount = 14,
       kHostM = kCtrCount, kHostWords };
// kCtrError: 1 a ray of more than 2^24 steps (reproducible mode), 2 a voxel beyond +-2^20 on a walk, 3 a sweep's tile never
// reported, and from the merged integrator's kernels:
enum { kErrMergedRayTooLong = 4, kErrMergedKeyCorner = 5 };


struct DetScratch;  // vgx_tsdf_det.hip
void det_scratch_free(DetScratch* s);

// The stable sorts of the sort-based paths.  A scan is 10^4 .. 10^6 records: below 2^20 items rocprim's radix
// sort is a merge sort -- blocks of 1024 sorted items merged pairwise, and above 201 072 items every pass is
// two launches (partition + merge path).  At these sizes a pass is a launch's latency, not bandwidth
// (profiles/r04_tsdf_launches.txt: 17 launches of 4-7 us for 237 568 records), so: 4096 items per sorted block
// (two passes fewer) and the one-launch odd-even merge all the way to 2^20 items (7 launches for the same
// records).  Above 2^20 items: rocprim's onesweep, as before.  VGX_TSDF_SORT=default switches back (A/B aid).
// (An own LSD radix sort -- one counting launch + one launch per 8 key bits, tiles chained by epoch-tagged words or
// by a count matrix -- was written, is correct, and is NOT faster: fewer launches, but three dependent memory
// round trips in each against the merge passes' two.  profiles/dropped/vgx_slot_sort.hip, profiles/README.md.)
using FewPassSort = rocprim::radix_sort_config<rocprim::default_config, rocprim::merge_sort_config<256, 512, 8, 128, 128, 4, (1u << 20)>,
                                               rocprim::default_config, (1u << 20)>;
inline bool few_pass_sort() {
  static const bool on = !(getenv("VGX_TSDF_SORT") && !strcmp(getenv("VGX_TSDF_SORT"), "default"));
  return on;
}
template <class KeyIn, class KeyOut, class ValIn, class ValOut>
inline hipError_t stable_sort_pairs(void* tmp, size_t& bytes, KeyIn keys_in, KeyOut keys_out, ValIn vals_in, ValOut vals_out,
                                    size_t n, unsigned end_bit, hipStream_t st) {
  if (few_pass_sort())
    return rocprim::radix_sort_pairs<FewPassSort>(tmp, bytes, keys_in, keys_out, vals_in, vals_out, n, 0, end_bit, st);
  return rocprim::radix_sort_pairs(tmp, bytes, keys_in, keys_out, vals_in, vals_out, n, 0, end_bit, st);
}
// 64-bit keys (the merged integrator's {clearing, end voxel} keys): the radix block sort of the configuration above
// walks all 64 bits (30 us for 65 536 records); a comparison sort does not care how wide the key is.  rocprim's
// merge_sort is stable; blocks of 1024 / 2048 records, odd-even merges (profiles/probes/sort_probe.hip: 65 536
// records 41.6 us against 57.7, 307 200 records 84.7 against 103.2).
using WideKeySortSmall = rocprim::merge_sort_config<256, 256, 4, 128, 128, 4, (1u << 20)>;
using WideKeySortLarge = rocprim::merge_sort_config<256, 256, 8, 128, 128, 4, (1u << 20)>;
inline hipError_t stable_sort_pairs_u64(void* tmp, size_t& bytes, const unsigned long long* keys_in, unsigned long long* keys_out,
                                        const unsigned int* vals_in, unsigned int* vals_out, size_t n, hipStream_t st) {
  if (!few_pass_sort() || n > (1u << 20))
    return rocprim::radix_sort_pairs(tmp, bytes, keys_in, keys_out, vals_in, vals_out, n, 0, 64, st);
  if (n > (1u << 17))
    return rocprim::merge_sort<WideKeySortLarge>(tmp, bytes, keys_in, keys_out, vals_in, vals_out, n, rocprim::less<unsigned long long>(), st);
  return rocprim::merge_sort<WideKeySortSmall>(tmp, bytes, keys_in, keys_out, vals_in, vals_out, n, rocprim::less<unsigned long long>(), st);
}

}  // namespace vgx

struct TsdfStats {  // one device allocation: TsdfLayerDev::n_blocks / ::dropped point into it
  int32_t n_blocks;
  int32_t pad;
  unsigned long long dropped;
};

struct vgx_tsdf_layer_s {
  vgx_ctx ctx = nullptr;
  vgx::TsdfLayerDev dev{};
  size_t lut_cells = 0;
  TsdfStats* d_stats = nullptr;
  // What the host knows about the device's allocation counter without waiting for it: the value
  // as of scan `known_seq` (read back asynchronously after scans) plus an upper bound on what the
  // scans launched since may have allocated.  known + pending is never below the true count.
  TsdfStats* h_stats = nullptr;  // pinned
  hipEvent_t readback_done = nullptr;
  bool readback_inflight = false;
  uint64_t scan_seq = 0, inflight_seq = 0, known_seq = 0;
  int64_t known_blocks = 0;
  std::vector<std::pair<uint64_t, int64_t>> recent;  // (scan, bound) of scans after known_seq
  unsigned long long dropped_seen = 0;
  int64_t growths = 0;  // re-boxings + pool enlargements so far
  float bound_reach = -1.0f;  // blocks a scan of this reach can touch (computed once per reach)
  int64_t bound_blocks = 0;
};

struct vgx_tsdf_integrator_s {
  vgx_ctx ctx = nullptr;
  vgx_tsdf_layer layer = nullptr;
  vgx::TsdfIntegratorDev dev{};
  long long reset_counter = 0;
  float* d_points = nullptr;  // staging for host-pointer scans
  uint32_t* d_rgba = nullptr;
  long long staging_cap = 0;
  // ... and its host side: two pinned buffers ([n x 12 B points][n x 4 B colours]) filled in turn, so that the caller's
  // (pageable) arrays are consumed when vgx_tsdf_integrate returns while the upload and the scan run behind it; an event per
  // buffer says when its upload has finished and the buffer may be filled again
  char* h_stage[2] = {nullptr, nullptr};
  hipEvent_t stage_uploaded[2] = {nullptr, nullptr};
  long long h_stage_cap = 0;
  int stage_turn = 0;
  std::mutex mu;  // one scan at a time per integrator: the staging buffers belong to the scan in flight
  // MergedTsdfIntegrator scratch (grown on demand): sort keys / point indices (double-buffered),
  // group starts, {groups, surface entries} counters, radix-sort workspace
  unsigned long long* d_mkeys[2] = {nullptr, nullptr};
  unsigned int* d_midx[2] = {nullptr, nullptr};
  unsigned int* d_mstart = nullptr;
  unsigned int* d_mrank = nullptr;     // rank of every sorted entry's group
  unsigned int* d_mcounters = nullptr; // {groups, surface entries, surface groups, valid entries, a ray too long}
  float4* d_gpg = nullptr;             // per group: merged point (layer frame) + merged weight
  uint32_t* d_gcolor = nullptr;
  uint32_t* d_gflags = nullptr;
  uint32_t* d_gcount = nullptr;        // voxels on the group's ray
  void* d_msort = nullptr;
  size_t msort_bytes = 0;
  long long merged_cap = 0;
  unsigned long long* d_wg_stats = nullptr;  // counted racing scans: one row per workgroup (grown on demand)
  long long wg_stats_cap = 0, wg_stats_rows = 0;
  int cloud_width = 0;  // vgx_tsdf_integrator_set_cloud_width: points per row of the scans to come (0: unorganised)
  // the racing scan's launcher: vgx::launch_racing_scan unless the diagnostics library installed its event-logging twin
  // (same kernel template, TRACE = true; include/voxgraph_amd_bench.h vgx_tsdf_integrator_set_event_trace)
  hipError_t (*racing_launch)(hipStream_t, const vgx::TsdfLayerDev&, const vgx::TsdfIntegratorDev&, const float*, const float*,
                              const uint32_t*, long long, int, bool, int) = nullptr;
  unsigned long long* d_trace = nullptr;   // owned by the integrator once set (freed with it)
  bool counted_on_ctx = false;             // Context::tsdf_integrators holds this one
  vgx::DetScratch* det = nullptr;  // reproducible mode's buffers (vgx_tsdf_det.hip), grown on demand
  // reproducible mode, bounded speculation (vgx_tsdf_det.hip det_count_kernel): a scan whose complete walks are more
  // than det_cap_threshold steps is written out det_cap steps deep at first.  Nothing but time depends on either;
  // vgx_tsdf_integrator_set_speculation (bench header) lets the tests drive the extension logic on small scans.
  uint32_t det_cap = 32;
  uint32_t det_cap_threshold = 8u << 20;  // (a city LiDAR scan's 5 M steps: 1.54 ms swept as they are, 2.0-2.1 ms cut to 32 and extended twice;
                                          //  a depth image's 18 M: cut.  profiles/probes/pipeline_det.sh)
  // merged integrator: lanes per group in merged_merge_kernel (4 / 8 / 16), chosen from the previous scan's points per
  // group -- a LiDAR scan's groups hold one or two points, a depth image's five to ten.  Results do not depend on it.
  int merged_lanes = 4;
  // integration_order "sorted": squared-norm keys / point indices (double-buffered) + radix-sort workspace
  uint32_t* d_okey[2] = {nullptr, nullptr};
  uint32_t* d_oidx[2] = {nullptr, nullptr};
  void* d_osort = nullptr;
  size_t osort_bytes = 0;
  long long order_cap = 0;
};

namespace vgx {
// vgx_tsdf.hip: room for a scan (block table box + pool), counters read-back, ApproxHashSet reset
int tsdf_reserve_for_scan(vgx_tsdf_layer L, const float origin[3], float reach);
int64_t tsdf_last_scan_bound(vgx_tsdf_layer L);
void tsdf_request_readback(vgx_tsdf_layer L);
// vgx_tsdf_coop.hip: the racing scan (one workgroup per 256 points: start set, cooperative walk, per-voxel folds);
// stats: gather vgx_tsdf_integrator_walk_stats' numbers (a counted scan: I.wg_stats must hold a row per workgroup)
// cloud_width: points per row of an organised cloud (0: unorganised) -- only which workgroup takes which point
hipError_t launch_racing_scan(hipStream_t stream, const TsdfLayerDev& L, const TsdfIntegratorDev& I, const float T[7],
                              const float* d_points, const uint32_t* d_rgba, long long n, int freespace, bool stats,
                              int cloud_width);
inline long long racing_scan_workgroups(long long n, int cloud_width) {
  if (cloud_width > 0 && n % cloud_width == 0)
    return (long long)((cloud_width + 15) / 16) * ((n / cloud_width + 15) / 16);
  return (n + 255) / 256;
}
// vgx_tsdf_det.hip: one scan in the reproducible mode (vgx_tsdf_config.deterministic); the caller
// holds the integrator's and the context's locks, the approximate sets have been reset for the scan
// `order`: order[seq] = index of the point visited seq-th (integration_order "sorted"), nullptr = "mixed"
int det_integrate(vgx_tsdf_integrator I, const float T[7], const void* d_points, const void* d_rgba, int64_t n,
                  int32_t freespace, const uint32_t* order, int64_t* n_updates);
// the scan's counters (kCtr*) for the merged integrator's own kernels; makes the scratch object on demand
int det_counters(vgx_tsdf_integrator I, unsigned long long** d_ctr);
int det_next_chain(vgx_tsdf_integrator I, uint32_t tiles, TileChain* chain);
// the merged integrator's rays (merged point / colour / flags / ray length per group, groups in key order)
// applied voxel by voxel in group order
int det_merged_commit(vgx_tsdf_integrator I, const float T[7], long long n, const float4* g_pg, const uint32_t* g_color,
                      const uint32_t* g_flags, uint32_t* g_count, const unsigned long long* keys_sorted,
                      const unsigned int* group_start, const unsigned int* counters, int64_t* n_updates);
}  // namespace vgx

#endif  // VGX_TSDF_INTERNAL_H_
