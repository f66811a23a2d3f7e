// TSDF path, REPRODUCIBLE mode (vgx_tsdf_config.deterministic): voxblox::FastTsdfIntegrator::
// integratePointCloud as it runs with integrator_threads = 1 in either integration_order_mode -- "mixed"
// (computed in the kernels) or "sorted" (a table built by a stable radix sort, vgx_tsdf.hip visiting_order)
// (the orders oracle/tsdf_oracle.c restates; call site voxgraph/src/frontend/measurement_processors/
// pointcloud_integrator.cpp:83), resolved in parallel.
//
// What makes the integrator order dependent is (a) the two approximate hash sets -- whether a ray is
// cast at all, and where it stops (after > max_consecutive_ray_collisions already-observed voxels in
// a row), depends on what every EARLIER ray wrote into the set slots it touches -- and (b) the
// running weighted average of a voxel, clamped after every update.  Both are functions of the
// visiting order alone, so they can be evaluated without walking it one ray at a time:
//
//  1. start set.  Every valid point exchanges its start cell's hash unconditionally, so "already
//     present" = "the previous point, in visiting order, that hit the same slot wrote the same
//     value": one stable radix sort by slot, one look at the left neighbour.
//  2. observed set.  Every cast ray's COMPLETE walk is written out (speculation: nobody stops
//     early), (slot, ray, step) stable-sorted by slot.  Given a stopping step T_r per ray, an
//     access happens iff step <= T_r, and what an access finds in its slot is the value of the last
//     access that HAPPENED before it in the slot's run: an exclusive max-scan over "position if
//     happened" gives that for all accesses at once (det_sweep_kernel: tiles chained by one tagged
//     word each, the scan itself never stored); every ray then re-reads its own flags and
//     recomputes T_r (det_ray_kernel, whose last workgroup tells the host through a pinned word).
//     Iterated from T_r = full length this is a fixed-point iteration whose
//     unique fixed point is the sequential execution (by induction over the visiting order: the
//     first ray whose T is wrong has only correct predecessors and is corrected by the next
//     sweep), reached when a sweep changes no T.  Measured: 15-25 sweeps on dense LiDAR scans
//     (the set of unsettled rays roughly halves every two sweeps).
//  3. voxel updates.  The accesses that happened and did not stop their ray are compacted in sorted
//     order: all updates of a voxel are then contiguous (same slot) and in visiting order.  New
//     blocks are allocated in the order of their first update in visiting order (as a sequential
//     run allocates them), and one thread per slot run applies its updates one after another with
//     plain loads and stores -- updateTsdfVoxel's arithmetic, no atomics.
//
// HBM traffic is a few sorts and scans over the speculative accesses (~60 B each); what the mode costs is
// launches and host waits (44 launches and two waits outside the sweeps per LiDAR scan): it
// trades 5-6 x the racing kernel's time for a layer that is the same bit for bit on every run.
#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include <rocprim/rocprim.hpp>

#include "vgx_tsdf_internal.h"

#pragma clang fp contract(off)

namespace vgx {

namespace {

struct Buf {
  void* p = nullptr;
  size_t bytes = 0;
  template <class T>
  T* as() const { return (T*)p; }
};

}  // namespace

struct DetScratch {
  // per point, indexed by position in the visiting order
  Buf ray_pg, ray_color, ray_flags, start_val, start_key, start_key_sorted, start_seq_sorted, count, off, T, broke,
      full_count, ext, walked;
  // per speculative access
  Buf acc_vox, acc_key, acc_ray, s_key, s_idx, s_r, s_k, s_h, last, seen, c_idx, c_key;
  Buf tmp;  // rocprim temporary storage
  // block allocation
  Buf first_touch, new_cells, new_cells_sorted, new_keys, new_keys_sorted, long_runs, t_at, t_sdf, t_w, t_color, t_far;
  bool first_touch_dirty = false;  // a scan failed between marking and assigning: refill
  // device counters (u64 each, enum below) + a pinned mirror; behind the mirror, pinned too, the other small
  // values a scan reads back (kHost*): every copy of a read-back queues behind the previous one and only the
  // last synchronise waits (a copy into pageable memory is a stream round trip of its own)
  unsigned long long* d_ctr = nullptr;
  unsigned long long* h_ctr = nullptr;
  // the sweeps (det_sweep_kernel): one 8-byte state per tile of the sorted accesses, tagged with the sweep's
  // epoch so that nothing has to be cleared between sweeps; zeroed when (re)allocated
  Buf tile_state;
  uint32_t sweep_epoch = 0;      // ... and of every TileChain launch (one counter: the words are shared)
  uint32_t chain_tickets = 0;    // TileChain workgroups launched since the scan's first kernel zeroed kCtrChainTicket
  // what the sweep's last block tells the host (pinned, written from the kernel): sweep sequence number << 2 |
  // error << 1 | "a stopping step moved"
  unsigned long long* h_flag = nullptr;
  unsigned long long* d_flag = nullptr;  // the same word as the device addresses it
  unsigned long long flag_seq = 0;
  // the counters' way to the host (read_counters): a one-workgroup kernel stores them into pinned, coherent words and
  // then a sequence number, the host polls that number -- no copy engine, no stream synchronisation (round 5)
  unsigned long long* h_report = nullptr;
  unsigned long long* d_report = nullptr;  // the same words as the device addresses them
  unsigned long long report_seq = 0;
  bool start_capped = false;  // the previous scan's complete walks were too many to write out: count capped at once
  long long ext_points = -1;  // ext[] holds the previous scan's marks for a scan of this many points (-1: nothing to keep)
};

enum { kRayValid = 1u, kRayClearing = 2u, kRayCast = 4u };
constexpr uint32_t kInvalidStartKey = 1u << kSetBits;
constexpr long long kVoxBias = 1ll << 20;  // 21 bits per axis

void det_scratch_free(DetScratch* s) {
  if (!s) return;
  Buf* all[] = {&s->ray_pg, &s->ray_color, &s->ray_flags, &s->start_val, &s->start_key, &s->start_key_sorted,
                &s->start_seq_sorted, &s->count, &s->off, &s->T, &s->broke, &s->full_count, &s->ext, &s->walked, &s->acc_vox, &s->acc_key, &s->acc_ray,
                &s->s_key, &s->s_idx, &s->s_r, &s->s_k, &s->s_h, &s->last, &s->seen, &s->c_idx, &s->c_key, &s->tmp,
                &s->first_touch, &s->new_cells, &s->new_cells_sorted, &s->new_keys, &s->new_keys_sorted, &s->long_runs,
                &s->t_at, &s->t_sdf, &s->t_w, &s->t_color, &s->t_far, &s->tile_state};
  for (Buf* b : all)
    if (b->p) (void)hipFree(b->p);
  if (s->d_ctr) (void)hipFree(s->d_ctr);
  if (s->h_ctr) (void)hipHostFree(s->h_ctr);
  if (s->h_flag) (void)hipHostFree(s->h_flag);
  if (s->h_report) (void)hipHostFree(s->h_report);
  delete s;
}

namespace {

__device__ __forceinline__ unsigned long long pack_vox(int x, int y, int z) {
  return (((unsigned long long)(x + kVoxBias) & 0x1fffffull) << 42) | (((unsigned long long)(y + kVoxBias) & 0x1fffffull) << 21) |
         ((unsigned long long)(z + kVoxBias) & 0x1fffffull);
}
__device__ __forceinline__ void unpack_vox(unsigned long long v, int& x, int& y, int& z) {
  x = (int)((long long)((v >> 42) & 0x1fffffull) - kVoxBias);
  y = (int)((long long)((v >> 21) & 0x1fffffull) - kVoxBias);
  z = (int)((long long)(v & 0x1fffffull) - kVoxBias);
}
// LongIndexHash [recalled]
__device__ __forceinline__ unsigned int index_hash(int x, int y, int z) {
  return (unsigned int)x + (unsigned int)y * 17191u + (unsigned int)z * 295530481u;
}

// ---- 1. points in visiting order: validity, transform, weight, start cell ------------------------
__global__ __launch_bounds__(256) void det_points_kernel(vgx_tsdf_config c, float vsi, float qw, float qx, float qy,
                                                        float qz, float tx, float ty, float tz,
                                                        const float* __restrict__ points_C,
                                                        const uint32_t* __restrict__ rgba, long long n,
                                                        const uint32_t* __restrict__ order,
                                                        int freespace_points, unsigned long long start_offset,
                                                        float4* __restrict__ ray_pg, uint32_t* __restrict__ ray_color,
                                                        uint32_t* __restrict__ ray_flags,
                                                        unsigned long long* __restrict__ start_val,
                                                        uint32_t* __restrict__ start_key, uint8_t* __restrict__ ext,
                                                        int ext_init, unsigned long long* __restrict__ ctr) {
  const long long seq = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  // this scan's counters and "written out completely" marks start here (the scan's first kernel: two fills saved)
  if (seq == 0) {
#pragma unroll
    for (int i = 0; i < kCtrCount; ++i) ctr[i] = 0ull;  // (one thread: a scan may have fewer points than counters)
  }
  if (seq <= n && ext_init < 2) ext[seq] = (uint8_t)ext_init;  // (2: the previous scan's marks stay, det_extend_kernel)
  if (seq >= n) return;
  const long long pi = visiting_order_point(order, seq, n);
  const float px = points_C[3 * pi], py = points_C[3 * pi + 1], pz = points_C[3 * pi + 2];
  // isPointValid
  bool valid = true, is_clearing = false;
  const float ray_distance = norm3(px, py, pz);
  if (ray_distance < c.min_ray_length_m) {
    valid = false;
  } else if (ray_distance > c.max_ray_length_m) {
    if (c.allow_clear || freespace_points) is_clearing = true; else valid = false;
  } else {
    is_clearing = freespace_points != 0;
  }
  if (!valid) {
    ray_flags[seq] = 0u;
    start_key[seq] = kInvalidStartKey;
    return;
  }
  float gx, gy, gz;
  transform_point(qw, qx, qy, qz, tx, ty, tz, px, py, pz, gx, gy, gz);
  // getVoxelWeight
  float weight = 1.0f;
  if (!c.use_const_weight) {
    const float dist_z = fabsf(pz);
    weight = dist_z > 1e-6f ? 1.0f / (dist_z * dist_z) : 0.0f;
  }
  const float sub_inv = c.start_voxel_subsampling_factor * vsi;
  const int sx = grid_index(gx * sub_inv + 1e-6f), sy = grid_index(gy * sub_inv + 1e-6f),
            sz = grid_index(gz * sub_inv + 1e-6f);
  const unsigned long long v = (unsigned long long)index_hash(sx, sy, sz) + start_offset;
  ray_pg[seq] = make_float4(gx, gy, gz, weight);
  ray_color[seq] = rgba ? rgba[pi] : 0u;
  ray_flags[seq] = kRayValid | (is_clearing ? kRayClearing : 0u);
  start_val[seq] = v;
  start_key[seq] = (uint32_t)(v & kSetMask);
}

// ---- start set: "already present" = the previous point that hit this slot wrote the same value ----
// Which rays are cast.  (What the slot holds afterwards -- what its last point wrote -- is stored by
// start_set_after, from the next kernel: every thread of this one has to have read the old value first.)
__global__ __launch_bounds__(256) void det_start_kernel(long long n, const uint32_t* __restrict__ key_sorted,
                                                       const uint32_t* __restrict__ seq_sorted,
                                                       const unsigned long long* __restrict__ start_val,
                                                       const unsigned long long* __restrict__ start_set,
                                                       uint32_t* __restrict__ ray_flags) {
  const long long p = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (p >= n) return;
  const uint32_t key = key_sorted[p];
  if (key == kInvalidStartKey) return;
  const uint32_t seq = seq_sorted[p];
  const unsigned long long mine = start_val[seq];
  const unsigned long long prev = (p > 0 && key_sorted[p - 1] == key) ? start_val[seq_sorted[p - 1]] : start_set[key];
  if (prev != mine) ray_flags[seq] |= kRayCast;
}
__device__ __forceinline__ void start_set_after(long long p, long long n, const uint32_t* __restrict__ key_sorted,
                                                const uint32_t* __restrict__ seq_sorted,
                                                const unsigned long long* __restrict__ start_val,
                                                unsigned long long* __restrict__ start_set) {
  if (p >= n) return;
  const uint32_t key = key_sorted[p];
  if (key != kInvalidStartKey && (p == n - 1 || key_sorted[p + 1] != key)) start_set[key] = start_val[seq_sorted[p]];
}

// ---- 2a. how far every cast ray is written out --------------------------------------------------------
// Speculating every ray's COMPLETE walk writes out 10-30 x what happens: all but a few "pioneer" rays of
// a scan stop within a handful of voxels (their neighbours have just been there).  So a ray is written
// out `cap` steps deep at first; rays that reach the end of what was written without having stopped
// (ext[] set by det_extend_kernel) get their complete walk in the next attempt, the others keep theirs.
// The fixed point of an attempt in which no ray is cut short is the fixed point of the complete system
// (same accesses happen, same decisions; it is unique), so the result does not depend on `cap`.
// (the depth is vgx_tsdf_integrator_s::det_cap, 32 unless a test says otherwise)

// (also the rays' offsets into the write-out: the exclusive prefix of the counts, taken inside this launch -- TileChain)
__global__ __launch_bounds__(256) void det_count_kernel(vgx_tsdf_config c, float vsi, float tx, float ty, float tz,
                                                       long long n, const float4* __restrict__ ray_pg,
                                                       const uint32_t* __restrict__ ray_flags,
                                                       const uint8_t* __restrict__ ext, uint32_t* __restrict__ count,
                                                       uint32_t* __restrict__ full_count,
                                                       const uint32_t* __restrict__ key_sorted,
                                                       const uint32_t* __restrict__ seq_sorted,
                                                       const unsigned long long* __restrict__ start_val,
                                                       unsigned long long* __restrict__ start_set, uint32_t cap,
                                                       unsigned long long* __restrict__ ctr, TileChain chain,
                                                       uint32_t* __restrict__ off) {
  __shared__ uint32_t sh_word, sh4[4];
  const uint32_t tile = chain_tile(chain, &sh_word);
  const long long seq = (long long)tile * 256 + threadIdx.x;
  // (a second job for the same thread index: the start set's state after this scan; a repeated count stores the
  // same values again)
  start_set_after(seq, n, key_sorted, seq_sorted, start_val, start_set);
  uint32_t full = 0, written = 0;
  if (seq < n && (ray_flags[seq] & kRayCast)) {
    const float4 g = ray_pg[seq];
    const RayDda r = ray_setup(c, vsi, tx, ty, tz, g.x, g.y, g.z, (ray_flags[seq] & kRayClearing) != 0, false);
    if (!r.bad) {
      if (r.steps + 1 >= (1ll << 24)) ctr[kCtrError] = 1ull;  // step index is packed into 24 bits below
      else full = (uint32_t)(r.steps + 1);
    }
  }
  if (seq <= n) {
    const bool complete = seq < n && ext[seq] != 0;
    written = complete ? full : min(full, cap);
    count[seq] = written;  // count[n] = 0
    if (seq < n) full_count[seq] = full;
  }
  uint32_t in_tile = 0;
  const uint32_t before = block_exclusive_sum(written, sh4, in_tile);
  const uint32_t prefix = chain_exclusive_sum(chain, tile, in_tile, &sh_word);
  if (seq <= n) off[seq] = prefix + before;  // (32 bits: a scan of 2^32 steps or more is refused by the 64-bit total below)
  // the scan's total, in 64 bits: one atomic per workgroup
  if (threadIdx.x == 0 && in_tile) atomicAdd(&ctr[kCtrTotal], (unsigned long long)in_tile);
}

// After the sweeps of an attempt: rays that did not stop within what was written out of them have to be written
// out completely (overflow: another attempt).  The marks also stay for the NEXT scan of as many points: the rays
// that run on are the "pioneers" of their neighbourhood in the visiting order, and a sensor that moves a few
// centimetres between scans has the same pioneers -- so the next scan writes them out completely at once and, as
// a rule, settles in one attempt (measured on the bench's depth camera: 2 attempts -> 1 for three scans in four).
// A mark is dropped again when its ray stopped well inside the cap after all.  Nothing depends on the marks but time.
__global__ __launch_bounds__(256) void det_extend_kernel(long long n, const uint32_t* __restrict__ count,
                                                        const uint32_t* __restrict__ full_count,
                                                        const uint8_t* __restrict__ broke, const int32_t* __restrict__ T,
                                                        uint8_t* __restrict__ ext, uint32_t cap, int mark_life,
                                                        unsigned long long* __restrict__ ctr) {
  const long long seq = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (seq == 0) ctr[kCtrTotal] = 0ull;  // (the next attempt counts again)
  if (seq >= n) return;
  const bool ran_on = count[seq] < full_count[seq] && !broke[seq];
  // (half the cap: a ray that got that far is as good as a pioneer for the next scan; a mark outlives a few scans in
  // which its ray stopped early: the pioneers of a moving sensor come and go)
  const bool far = full_count[seq] > cap && T[seq] >= (int32_t)(cap / 2);
  const uint8_t old = ext[seq];
  ext[seq] = (ran_on || far) ? (uint8_t)mark_life : (old > 0 ? old - 1 : 0);
  if (ran_on) ctr[kCtrOverflow] = 1ull;
}

// ---- 2b. the complete walks, written out ---------------------------------------------------------
__global__ __launch_bounds__(256) void det_walk_kernel(vgx_tsdf_config c, float vsi, float tx, float ty, float tz,
                                                      long long n, const float4* __restrict__ ray_pg,
                                                      const uint32_t* __restrict__ ray_flags,
                                                      const uint32_t* __restrict__ count,
                                                      const uint32_t* __restrict__ off,
                                                      unsigned long long observed_offset,
                                                      unsigned long long* __restrict__ acc_vox,
                                                      uint32_t* __restrict__ acc_key, uint32_t* __restrict__ acc_ray,
                                                      int32_t* __restrict__ T, uint8_t* __restrict__ broke,
                                                      uint32_t* __restrict__ walked, int warm,
                                                      unsigned long long* __restrict__ ctr) {
  const long long seq = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (seq == 0) ctr[kCtrOverflow] = 0ull;  // (what det_extend_kernel reports at the end of this attempt)
  if (seq >= n) return;
  const uint32_t cnt = count[seq];
  // Speculation: nobody stops early.  In a later attempt of the same scan (`warm`) a ray that is written out as far
  // as before starts from where the previous attempt left it: the iteration reaches its one fixed point from any
  // start (a ray's first wrong step is corrected by the next sweep once the rays before it are right), and most
  // rays are not touched by what the few extended ones add.
  if (!(warm && walked[seq] == cnt)) {
    T[seq] = (int32_t)cnt - 1;
    broke[seq] = 0;
  }
  walked[seq] = cnt;
  if (cnt == 0) return;
  const float4 g = ray_pg[seq];
  RayDda r = ray_setup(c, vsi, tx, ty, tz, g.x, g.y, g.z, (ray_flags[seq] & kRayClearing) != 0, false);
  const size_t base = off[seq];
  bool out_of_range = false;
  for (uint32_t k = 0; k < cnt; ++k) {
    const int vx = r.curr[0], vy = r.curr[1], vz = r.curr[2];
    dda_advance(r);
    out_of_range |= vx <= -kVoxBias || vx >= kVoxBias || vy <= -kVoxBias || vy >= kVoxBias || vz <= -kVoxBias || vz >= kVoxBias;
    acc_vox[base + k] = pack_vox(vx, vy, vz);
    acc_key[base + k] = (uint32_t)(((unsigned long long)index_hash(vx, vy, vz) + observed_offset) & kSetMask);
    acc_ray[base + k] = (uint32_t)seq;
  }
  if (out_of_range) ctr[kCtrError] = 2ull;
}

// after the sort by slot: what the sweeps read, in sorted order
__global__ __launch_bounds__(256) void det_gather_kernel(size_t N, const uint32_t* __restrict__ s_idx,
                                                        const unsigned long long* __restrict__ acc_vox,
                                                        const uint32_t* __restrict__ acc_ray,
                                                        const uint32_t* __restrict__ off, uint32_t* __restrict__ s_r,
                                                        uint32_t* __restrict__ s_k, uint32_t* __restrict__ s_h) {
  const size_t p = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (p >= N) return;
  const uint32_t idx = s_idx[p];
  const uint32_t r = acc_ray[idx];
  int x, y, z;
  unpack_vox(acc_vox[idx], x, y, z);
  s_r[p] = r;
  s_k[p] = idx - off[r];
  s_h[p] = index_hash(x, y, z);
}

// scan inputs
struct HappenedOp {  // position + 1 of an access whose exchange happens, else 0
  const uint32_t* s_r;
  const uint32_t* s_k;
  const int32_t* T;
  __host__ __device__ uint32_t operator()(uint32_t p) const { return (int32_t)s_k[p] <= T[s_r[p]] ? p + 1u : 0u; }
};
struct UpdateOp {  // 1 for an access that updates its voxel
  const uint32_t* s_r;
  const uint32_t* s_k;
  const int32_t* T;
  const uint8_t* broke;
  const uint32_t* s_idx;   // with `flags` (merged integrator): the decision was taken when the ray was written out
  const uint8_t* flags;
  uint32_t N;
  __host__ __device__ uint32_t operator()(uint32_t p) const {
    if (p >= N) return 0u;
    if (flags) return flags[s_idx[p]] ? 1u : 0u;
    // fast integrator: the exchange happened and did not stop the ray
    const uint32_t r = s_r[p];
    const int32_t k = (int32_t)s_k[p], t = T[r];
    return (k < t || (k == t && !broke[r])) ? 1u : 0u;
  }
};

// ---- 2c. one sweep: what every access finds in its slot ... ---------------------------------------
// (the two-kernel form over a rocprim scan: kept for the 1 / 4 of the comparison in profiles/ -- VGX_DET_SWEEP=scan)
__global__ __launch_bounds__(256) void det_seen_kernel(size_t N, const uint32_t* __restrict__ s_key,
                                                      const uint32_t* __restrict__ s_idx,
                                                      const uint32_t* __restrict__ s_h,
                                                      const uint32_t* __restrict__ last,
                                                      const unsigned long long* __restrict__ observed_set,
                                                      unsigned long long observed_offset, uint8_t* __restrict__ seen) {
  const size_t p = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (p >= N) return;
  const uint32_t key = s_key[p], h = s_h[p], prev = last[p];
  bool present;
  if (prev > 0 && s_key[prev - 1] == key) present = s_h[prev - 1] == h;           // an earlier access of this scan
  else present = observed_set[key] == (unsigned long long)h + observed_offset;    // what earlier scans left
  seen[s_idx[p]] = present ? 1 : 0;
}

// The sweep in ONE launch: the exclusive max-scan of "position + 1 if the access happened" and what it is used
// for, without the scan's output ever reaching memory.  Tiles of 2048 consecutive sorted accesses, taken in
// the order the workgroups start (a ticket), chained by one 8-byte state per tile:
//     epoch << 34 | status << 32 | value       status 1: the tile's own last happened access (0: none),
//                                               status 2: the last happened access up to the tile's end
// Positions only grow, so "max" is "the nearest one before": a tile that holds a happened access publishes its
// status-2 word before it looks at anybody, and a look back ends at the first earlier tile whose value is not 0
// -- one step on everything but pathological input (a tile waits only for tiles that started before it, which
// never wait for it).  The words are the whole exchange between workgroups (relaxed agent-scope 8-byte loads
// and stores, tag and value in one word); the epoch is the sweep's number, so no word is ever cleared.
// Also written, for det_finish_kernel: `last` at the end of every slot run.
#ifndef VGX_SWEEP_IPT
#define VGX_SWEEP_IPT 8  // (A/B builds: make SUFFIX=_ipt16 EXTRA=-DVGX_SWEEP_IPT=16)
#endif
constexpr int kSweepIpt = VGX_SWEEP_IPT;
static_assert(kSweepIpt % 4 == 0, "whole 16-byte loads");
constexpr uint32_t kSweepTile = 256u * kSweepIpt;
constexpr unsigned long long kSweepEpochMax = (1ull << 30) - 1;

__device__ __forceinline__ void load8(const uint32_t* __restrict__ a, uint32_t base, uint32_t N, uint32_t (&v)[kSweepIpt]) {
  if (base + kSweepIpt <= N) {  // base is a multiple of kSweepIpt: aligned 16-byte loads
#pragma unroll
    for (int q = 0; q < kSweepIpt / 4; ++q) {
      const uint4 w = *(const uint4*)(a + base + 4 * q);
      v[4 * q] = w.x; v[4 * q + 1] = w.y; v[4 * q + 2] = w.z; v[4 * q + 3] = w.w;
    }
  } else {
#pragma unroll
    for (int e = 0; e < kSweepIpt; ++e) v[e] = base + e < N ? a[base + e] : 0u;
  }
}

__global__ __launch_bounds__(256) void det_sweep_kernel(uint32_t N, unsigned long long epoch,
                                                       unsigned long long* __restrict__ ctr,
                                                       unsigned long long* __restrict__ tile_state,
                                                       const uint32_t* __restrict__ s_key,
                                                       const uint32_t* __restrict__ s_idx,
                                                       const uint32_t* __restrict__ s_h,
                                                       const uint32_t* __restrict__ s_r,
                                                       const uint32_t* __restrict__ s_k, const int32_t* __restrict__ T,
                                                       const unsigned long long* __restrict__ observed_set,
                                                       unsigned long long observed_offset, uint8_t* __restrict__ seen,
                                                       uint32_t* __restrict__ last) {
  __shared__ uint32_t sh_tile, sh_prefix, sh_wave[4];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  if (tid == 0) sh_tile = (uint32_t)atomicAdd(&ctr[kCtrTicket], 1ull);
  __syncthreads();
  const uint32_t tile = sh_tile;
  const uint32_t base = tile * kSweepTile + (uint32_t)tid * kSweepIpt;
  // (every gather of a batch is issued before the first one is waited for: the loads are unconditional -- entries
  // beyond N read element 0 -- and what they return is selected afterwards)
  uint32_t r[kSweepIpt], k[kSweepIpt], hp[kSweepIpt];
  int32_t stop[kSweepIpt];
  load8(s_r, base, N, r);
  load8(s_k, base, N, k);
#pragma unroll
  for (int e = 0; e < kSweepIpt; ++e) stop[e] = T[r[e]];
  uint32_t own = 0;  // this thread's last happened access (position + 1)
#pragma unroll
  for (int e = 0; e < kSweepIpt; ++e) {
    const uint32_t p = base + e;
    hp[e] = (p < N && (int32_t)k[e] <= stop[e]) ? p + 1u : 0u;
    own = hp[e] ? hp[e] : own;
  }
  uint32_t inc = own;  // inclusive over the wave's threads
#pragma unroll
  for (int d = 1; d < 64; d <<= 1) {
    const uint32_t o = (uint32_t)__shfl_up((int)inc, d);
    if (lane >= d) inc = max(inc, o);
  }
  if (lane == 63) sh_wave[wave] = inc;
  uint32_t before = (uint32_t)__shfl_up((int)inc, 1);  // exclusive: the threads before this one, in the tile
  if (lane == 0) before = 0;
  __syncthreads();
  uint32_t tile_last = 0;
#pragma unroll
  for (int w = 0; w < 4; ++w) {
    if (w < wave) before = max(before, sh_wave[w]);
    tile_last = max(tile_last, sh_wave[w]);
  }
  if (tid == 0) {
    const unsigned long long tag = epoch << 34;
    unsigned long long* mine = tile_state + tile;
    uint32_t prefix = 0;
    if (tile_last != 0 || tile == 0) __hip_atomic_store(mine, tag | (2ull << 32) | tile_last, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    else __hip_atomic_store(mine, tag | (1ull << 32), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    for (uint32_t t = tile; t-- > 0;) {
      unsigned long long x;
      unsigned spins = 0;
      while (((x = __hip_atomic_load(tile_state + t, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) >> 34) != epoch) {
        __builtin_amdgcn_s_sleep(1);
        if (++spins > (1u << 22)) {  // (seconds: the tile before this one never ran -- internal error, not a hang)
          ctr[kCtrError] = 3ull;
          x = tag | (2ull << 32);
          break;
        }
      }
      prefix = (uint32_t)x;
      if (prefix != 0 || ((x >> 32) & 3ull) == 2ull) break;
    }
    if (tile_last == 0 && tile != 0) __hip_atomic_store(mine, tag | (2ull << 32) | prefix, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    sh_prefix = prefix;
  }
  __syncthreads();
  uint32_t run = before ? before : sh_prefix;  // the last happened access before this thread's first one
  uint32_t key[kSweepIpt], h[kSweepIpt], idx[kSweepIpt], prev[kSweepIpt], prev_key[kSweepIpt], prev_h[kSweepIpt];
  unsigned long long left[kSweepIpt];
  load8(s_key, base, N, key);
  load8(s_h, base, N, h);
  load8(s_idx, base, N, idx);
  const uint32_t after = base + kSweepIpt < N ? s_key[base + kSweepIpt] : 0xffffffffu;  // (keys are < 2^20)
#pragma unroll
  for (int e = 0; e < kSweepIpt; ++e) {
    prev[e] = run;
    run = hp[e] ? hp[e] : run;
  }
#pragma unroll
  for (int e = 0; e < kSweepIpt; ++e) {
    const uint32_t at = prev[e] ? prev[e] - 1u : 0u;
    prev_key[e] = s_key[at];
    prev_h[e] = s_h[at];
    left[e] = observed_set[key[e]];  // (sorted by key: neighbours read the same or the next entries)
  }
#pragma unroll
  for (int e = 0; e < kSweepIpt; ++e) {
    const uint32_t p = base + e;
    const bool earlier = prev[e] > 0 && prev_key[e] == key[e];  // an earlier access of this scan, or what earlier scans left
    const bool present = earlier ? prev_h[e] == h[e] : left[e] == (unsigned long long)h[e] + observed_offset;
    const uint32_t next_key = e + 1 < kSweepIpt ? key[e + 1] : after;
    if (p < N) {
      seen[idx[e]] = present ? 1 : 0;
      if (p == N - 1 || next_key != key[e]) last[p] = prev[e];  // end of a slot run: what det_finish_kernel looks at
    }
  }
}

// ---- ... and where every ray stops, given that -------------------------------------------------------
// The last block to finish reports: to the host through a pinned word (the host polls it instead of
// synchronising the stream: it can keep sweeps queued meanwhile), and it puts the two counters back.
__global__ __launch_bounds__(256) void det_ray_kernel(long long n, int max_collisions, const uint32_t* __restrict__ count,
                                                     const uint32_t* __restrict__ off, const uint8_t* __restrict__ seen,
                                                     int32_t* __restrict__ T, uint8_t* __restrict__ broke,
                                                     unsigned long long* __restrict__ ctr, unsigned long long seq_no,
                                                     unsigned long long* __restrict__ host_flag) {
  const long long seq = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  int moved = 0;
  const uint32_t cnt = seq < n ? count[seq] : 0u;
  if (cnt != 0) {
    // the ray's flags, four at a time: aligned words around its (byte-aligned) run, shifted into place -- a ray of a
    // city scan has 80 of them, and a dependent byte load each made this kernel as long as the sweep kernel's third
    const size_t at = off[seq];
    const uint32_t* words = (const uint32_t*)seen + (at >> 2);  // (seen comes from hipMalloc: 256-byte aligned)
    const int shift = (int)(at & 3u) * 8;
    int collisions = 0;
    int32_t t = (int32_t)cnt - 1;
    uint8_t b = 0;
    uint32_t lo = words[0];
    for (uint32_t k0 = 0; k0 < cnt && !b; k0 += 4) {
      const uint32_t hi = words[(k0 >> 2) + 1];  // (at most 4 bytes past the ray's last flag: the buffer has slack)
      uint32_t four = shift ? (lo >> shift) | (hi << (32 - shift)) : lo;
      lo = hi;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        if (b || k0 + j >= cnt) break;
        if (four & 0xffu) ++collisions; else collisions = 0;
        if (collisions > max_collisions) {
          t = (int32_t)(k0 + j);
          b = 1;
        }
        four >>= 8;
      }
    }
    if (t != T[seq] || b != broke[seq]) {
      T[seq] = t;
      broke[seq] = b;
      moved = 1;
    }
  }
  const int any = __syncthreads_or(moved);
  if (threadIdx.x == 0) {
    const unsigned long long add = 1ull + (any ? (1ull << 32) : 0ull);
    const unsigned long long old = atomicAdd(&ctr[kCtrArrive], add);
    if ((uint32_t)old == gridDim.x - 1u) {  // the sweep's last block
      const unsigned long long changed = ((old + add) >> 32) != 0ull ? 1ull : 0ull;
      const unsigned long long err = __hip_atomic_load(&ctr[kCtrError], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) ? 2ull : 0ull;
      __hip_atomic_store(&ctr[kCtrArrive], 0ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      __hip_atomic_store(&ctr[kCtrTicket], 0ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      __hip_atomic_store(&ctr[kCtrChanged], changed, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      __hip_atomic_store(host_flag, (seq_no << 2) | err | changed, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);  // (the word is all the host reads)
    }
  }
}

// ---- 3a. compaction of the updates (sorted order kept) + the set's state after the scan ---------
// (the compaction offsets -- the exclusive prefix of `update` -- are taken inside this launch: TileChain; tiles of
// 2048 sorted accesses, eight consecutive ones per thread; every batch of gathers is issued before the first one is
// waited for, as in det_sweep_kernel: entries beyond N read element 0.  *n_updates = how many accesses update their
// voxel.  FLAGS: the merged integrator's form of `update` -- the decision was taken when the ray was written out.)
template <bool FLAGS>
__global__ __launch_bounds__(256) void det_finish_kernel(uint32_t N, const uint32_t* __restrict__ s_key,
                                                        const uint32_t* __restrict__ s_idx,
                                                        const uint32_t* __restrict__ s_h,
                                                        const uint32_t* __restrict__ last, UpdateOp update, TileChain chain,
                                                        uint32_t* __restrict__ n_updates, uint32_t* __restrict__ c_idx,
                                                        uint32_t* __restrict__ c_key,
                                                        unsigned long long* __restrict__ observed_set,
                                                        unsigned long long observed_offset) {
  __shared__ uint32_t sh_word, sh4[4];
  const uint32_t tile = chain_tile(chain, &sh_word);
  const uint32_t base = tile * kSweepTile + (uint32_t)threadIdx.x * kSweepIpt;
  uint32_t idx[kSweepIpt], u[kSweepIpt], hp[kSweepIpt], mine = 0;
  load8(s_idx, base, N, idx);
  if (FLAGS) {
    uint8_t f[kSweepIpt];
#pragma unroll
    for (int e = 0; e < kSweepIpt; ++e) f[e] = update.flags[idx[e]];
#pragma unroll
    for (int e = 0; e < kSweepIpt; ++e) {
      u[e] = (base + e < N && f[e]) ? 1u : 0u;
      hp[e] = 0u;  // (no approximate sets: unused)
    }
  } else {
    uint32_t r[kSweepIpt], k[kSweepIpt];
    int32_t stop[kSweepIpt];
    uint8_t broke[kSweepIpt];
    load8(update.s_r, base, N, r);
    load8(update.s_k, base, N, k);
#pragma unroll
    for (int e = 0; e < kSweepIpt; ++e) {
      stop[e] = update.T[r[e]];
      broke[e] = update.broke[r[e]];
    }
#pragma unroll
    for (int e = 0; e < kSweepIpt; ++e) {
      const bool in = base + e < N;
      const int32_t step = (int32_t)k[e];
      // the exchange happened and did not stop the ray (UpdateOp); it happened at all (HappenedOp)
      u[e] = (in && (step < stop[e] || (step == stop[e] && !broke[e]))) ? 1u : 0u;
      hp[e] = (in && step <= stop[e]) ? base + e + 1u : 0u;
    }
  }
#pragma unroll
  for (int e = 0; e < kSweepIpt; ++e) mine += u[e];
  uint32_t in_tile = 0;
  const uint32_t before = block_exclusive_sum(mine, sh4, in_tile);
  uint32_t q = chain_exclusive_sum(chain, tile, in_tile, &sh_word) + before;
  uint32_t key[kSweepIpt];
  load8(s_key, base, N, key);
  // the slot keeps what the last exchange of its run wrote: at the end of every run, the run's last happened access
  uint32_t li[kSweepIpt], li_key[kSweepIpt], li_h[kSweepIpt];
  if (observed_set) {
    uint32_t before_p[kSweepIpt];
    load8(last, base, N, before_p);  // (written at run ends by the last sweep; elsewhere unused)
#pragma unroll
    for (int e = 0; e < kSweepIpt; ++e) li[e] = hp[e] ? hp[e] : before_p[e];
#pragma unroll
    for (int e = 0; e < kSweepIpt; ++e) {
      const uint32_t at = li[e] ? li[e] - 1u : 0u;
      li_key[e] = s_key[at < N ? at : 0u];
      li_h[e] = s_h[at < N ? at : 0u];
    }
  }
  const uint32_t after = base + kSweepIpt < N ? s_key[base + kSweepIpt] : 0xffffffffu;
#pragma unroll
  for (int e = 0; e < kSweepIpt; ++e) {
    const uint32_t p = base + e;
    if (p >= N) break;
    if (u[e]) {
      c_idx[q] = idx[e];
      c_key[q] = key[e];
      ++q;
    }
    if (p == N - 1) *n_updates = q;
    if (!observed_set) continue;  // merged integrator: no approximate sets
    const uint32_t next_key = e + 1 < kSweepIpt ? key[e + 1] : after;
    if ((p == N - 1 || next_key != key[e]) && li[e] > 0 && li_key[e] == key[e])
      observed_set[key[e]] = (unsigned long long)li_h[e] + observed_offset;
  }
}

// ---- 3b. blocks the updates need that the layer does not have yet ----------------------------------
__device__ __forceinline__ long long lut_cell(const TsdfLayerDev& L, int bx, int by, int bz) {
  const int rx = bx - L.lut_min[0], ry = by - L.lut_min[1], rz = bz - L.lut_min[2];
  if ((unsigned)rx >= (unsigned)L.lut_dim[0] || (unsigned)ry >= (unsigned)L.lut_dim[1] || (unsigned)rz >= (unsigned)L.lut_dim[2])
    return -1;
  return rx + (long long)L.lut_dim[0] * (ry + (long long)L.lut_dim[1] * rz);
}

// (M_dev: the number of updates where only the device knows it yet -- the launch covers an upper bound)
__global__ __launch_bounds__(256) void det_blocks_kernel(TsdfLayerDev L, size_t M, const uint32_t* __restrict__ M_dev,
                                                        const uint32_t* __restrict__ c_idx,
                                                        const unsigned long long* __restrict__ acc_vox,
                                                        const uint32_t* __restrict__ acc_ray,
                                                        const uint32_t* __restrict__ off,
                                                        unsigned long long* __restrict__ first_touch,
                                                        int32_t* __restrict__ new_cells, uint32_t new_cap,
                                                        unsigned long long* __restrict__ ctr) {
  const size_t q = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  const size_t m = M_dev ? (size_t)*M_dev : M;
  if (q == 0) {  // what the host reads back with the counters
    ctr[kCtrM] = (unsigned long long)m;
    ctr[kCtrBlocks] = (unsigned long long)(uint32_t)*L.n_blocks;
  }
  if (q >= m) return;
  const uint32_t idx = c_idx[q];
  int x, y, z;
  unpack_vox(acc_vox[idx], x, y, z);
  const long long cell = lut_cell(L, x >> L.vps_shift, y >> L.vps_shift, z >> L.vps_shift);
  if (cell < 0 || L.lut[cell] != -1) return;
  const uint32_t r = acc_ray[idx];
  const unsigned long long when = ((unsigned long long)r << 24) | (unsigned long long)(idx - off[r]);
  if (atomicMin(&first_touch[cell], when) == ~0ull) {  // exactly one update per new block sees the fresh entry
    const unsigned long long j = atomicAdd(&ctr[kCtrNew], 1ull);
    if (j < new_cap) new_cells[j] = (int32_t)cell;
  }
}

__global__ void det_new_keys_kernel(uint32_t n_new, const int32_t* __restrict__ new_cells,
                                    const unsigned long long* __restrict__ first_touch,
                                    unsigned long long* __restrict__ new_keys) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n_new) new_keys[i] = first_touch[new_cells[i]];
}

// new blocks take pool slots in the order of their first update in visiting order
__global__ void det_assign_kernel(TsdfLayerDev L, uint32_t n_new, int32_t base, const int32_t* __restrict__ cells_sorted,
                                  unsigned long long* __restrict__ first_touch) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n_new) return;
  const long long cell = cells_sorted[i];
  first_touch[cell] = ~0ull;
  const long long slot = (long long)base + i;
  if (slot < L.max_blocks) {
    const long long dx = L.lut_dim[0], dy = L.lut_dim[1];
    L.block_index[3 * slot + 0] = (int32_t)(cell % dx) + L.lut_min[0];
    L.block_index[3 * slot + 1] = (int32_t)((cell / dx) % dy) + L.lut_min[1];
    L.block_index[3 * slot + 2] = (int32_t)(cell / (dx * dy)) + L.lut_min[2];
    L.lut[cell] = (int32_t)slot;
  } else {
    L.lut[cell] = -3;  // pool exhausted (cannot happen after tsdf_reserve_for_scan)
  }
  if (i == 0) *L.n_blocks = (int32_t)min((long long)base + n_new, (long long)L.max_blocks);
}

// ---- 3c. updateTsdfVoxel in a fixed order -----------------------------------------------------------
// computeDistance + updateTsdfVoxel's weight [recalled]; the same operations as update_voxel (vgx_tsdf.hip)
struct UpdateTerm {
  float sdf, w;
};
__device__ __forceinline__ UpdateTerm update_term(const vgx_tsdf_config& c, float vs, float tx, float ty, float tz,
                                                  const float4& g, int vx, int vy, int vz) {
  const float trunc = c.default_truncation_distance;
  const float cx = ((float)vx + 0.5f) * vs, cy = ((float)vy + 0.5f) * vs, cz = ((float)vz + 0.5f) * vs;
  const float vvx = cx - tx, vvy = cy - ty, vvz = cz - tz;
  const float vpx = g.x - tx, vpy = g.y - ty, vpz = g.z - tz;
  const float dist_G = norm3(vpx, vpy, vpz);
  const float dot = (vvx * vpx + vvy * vpy) + vvz * vpz;
  const float dist_G_V = dot / dist_G;
  UpdateTerm u;
  u.sdf = dist_G - dist_G_V;
  u.w = g.w;
  if (c.use_weight_dropoff && u.sdf < -vs) {
    u.w = g.w * (trunc + u.sdf) / (trunc - vs);
    u.w = fmaxf(u.w, 0.0f);
  }
  if (c.use_sparsity_compensation_factor && fabsf(u.sdf) < trunc) u.w *= c.sparsity_compensation_factor;
  return u;
}

// the running average itself (the voxel mutex's critical section)
__device__ __forceinline__ void apply_term(const vgx_tsdf_config& c, float sdf, float uw, uint32_t color, float& d, float& w,
                                           uint32_t& col, bool& dirty) {
  const float trunc = c.default_truncation_distance;
  const float new_weight = w + uw;
  if (new_weight < 1e-6f) return;  // kFloatEpsilon
  const float new_sdf = (sdf * uw + d * w) / new_weight;
  if (fabsf(sdf) < trunc) col = blended_color(col, color, w, uw);
  d = (new_sdf > 0.0f) ? fminf(trunc, new_sdf) : fmaxf(-trunc, new_sdf);
  w = fminf(c.max_weight, new_weight);
  dirty = true;
}

template <bool ALLOCATE>
__device__ __forceinline__ int block_slot(const TsdfLayerDev& L, int bx, int by, int bz) {
  if (ALLOCATE) return get_or_allocate_block(L, bx, by, bz);
  const long long cell = lut_cell(L, bx, by, bz);
  return cell >= 0 ? L.lut[cell] : -1;
}

constexpr int kShortRun = 32;

// What an update contributes, evaluated for all M updates in parallel (sorted order): the voxel it
// goes to, its sdf, its weight, its colour.  Only the running average itself is order dependent.
// ALLOCATE: blocks are taken from the pool on demand (merged integrator outside the reproducible mode:
// the values are order-exact all the same, only the pool order is arrival order); otherwise they have
// been allocated, in order, beforehand.
template <bool ALLOCATE>
__global__ __launch_bounds__(256) void det_terms_kernel(TsdfLayerDev L, vgx_tsdf_config c, float tx, float ty, float tz,
                                                       size_t M, const uint32_t* __restrict__ c_idx,
                                                       const unsigned long long* __restrict__ acc_vox,
                                                       const uint32_t* __restrict__ acc_ray,
                                                       const float4* __restrict__ ray_pg,
                                                       const uint32_t* __restrict__ ray_color,
                                                       long long* __restrict__ t_at, float* __restrict__ t_sdf,
                                                       float* __restrict__ t_w, uint32_t* __restrict__ t_color,
                                                       uint8_t* __restrict__ t_far) {
  const size_t q = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (q >= M) return;
  const uint32_t idx = c_idx[q];
  int vx, vy, vz;
  unpack_vox(acc_vox[idx], vx, vy, vz);
  const int vps = L.vps, shift = L.vps_shift, mask = vps - 1;
  const int slot = block_slot<ALLOCATE>(L, vx >> shift, vy >> shift, vz >> shift);
  const uint32_t r = acc_ray[idx];
  const UpdateTerm u = update_term(c, L.voxel_size, tx, ty, tz, ray_pg[r], vx, vy, vz);
  t_at[q] = slot < 0 ? -1ll
                     : (long long)((size_t)slot * ((size_t)vps * vps * vps) +
                                   (size_t)((vx & mask) + vps * ((vy & mask) + vps * (vz & mask))));
  t_sdf[q] = u.sdf;
  t_w[q] = u.w;
  t_color[q] = ray_color[r];
  // "cannot move a voxel that sits at +truncation and weighs at most max_weight" (det_apply_long_kernel)
  t_far[q] = (u.sdf >= 2.0f * c.default_truncation_distance && u.w * 1.0e5f >= c.max_weight && u.w >= 1e-6f) ? 1 : 0;
}

// One thread per slot run: all updates of a voxel are contiguous and in order.  Runs longer than
// kShortRun (voxels every ray crosses: the sensor's own neighbourhood under the merged integrator) are
// left to det_apply_long_kernel.
__global__ __launch_bounds__(256) void det_apply_kernel(TsdfLayerDev L, vgx_tsdf_config c, size_t M,
                                                       const uint32_t* __restrict__ c_key,
                                                       const long long* __restrict__ t_at,
                                                       const float* __restrict__ t_sdf, const float* __restrict__ t_w,
                                                       const uint32_t* __restrict__ t_color,
                                                       uint32_t* __restrict__ long_runs,
                                                       unsigned long long* __restrict__ ctr) {
  const size_t q0 = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (q0 >= M) return;
  const uint32_t key = c_key[q0];
  if (q0 > 0 && c_key[q0 - 1] == key) return;  // not the head of its run
  size_t q1 = q0 + 1;
  while (q1 < M && q1 - q0 <= (size_t)kShortRun && c_key[q1] == key) ++q1;
  if (q1 - q0 > (size_t)kShortRun) {
    long_runs[atomicAdd(&ctr[kCtrLong], 1ull)] = (uint32_t)q0;
    return;
  }
  long long at_cached = -1;
  float d = 0.0f, w = 0.0f;
  uint32_t col = 0u;
  bool dirty = false;
  unsigned long long dropped = 0;
  for (size_t q = q0; q < q1; ++q) {
    const long long at = t_at[q];
    if (at < 0) {
      ++dropped;
      continue;
    }
    if (at != at_cached) {  // another voxel that shares the slot: write the current one back
      if (dirty) {
        L.voxels[at_cached] = pack_voxel(d, w);
        L.rgba[at_cached] = col;
      }
      const unsigned long long v = L.voxels[at];
      d = __uint_as_float((unsigned)(v & 0xffffffffull));
      w = __uint_as_float((unsigned)(v >> 32));
      col = L.rgba[at];
      at_cached = at;
      dirty = false;
    }
    apply_term(c, t_sdf[q], t_w[q], t_color[q], d, w, col, dirty);
  }
  if (dirty) {
    L.voxels[at_cached] = pack_voxel(d, w);
    L.rgba[at_cached] = col;
  }
  if (dropped) {
    atomicAdd(L.dropped, dropped);
    atomicAdd(&ctr[kCtrDropped], dropped);
  }
}

// Long runs: one wavefront per run streams the run's terms 64 at a time (the next 64 are in flight
// while the current ones are applied) and runs the clamped running average -- a handful of dependent
// f32 operations per update -- over lane broadcasts.
// Updates that cannot move a voxel that already sits at +truncation only add their weight: with
// d = t, sdf >= 2 t and w >= 1e-5 W,  fl((sdf w + t W) / (W + w)) >= t [1 + w / (W + w)] (1 - 4 eps) >= t,
// so the clamp returns t exactly (eps = 2^-24; w / (W + w) >= 1e-5 >> 4 eps).  det_terms_kernel marks
// the updates for which that holds whatever the voxel weighs (sdf >= 2 t, 1e5 w >= max_weight >= W);
// where a whole sub-batch of 64 is marked (the sensor's own neighbourhood: free space, seen by every
// ray) one ballot turns it into a chain of 64 additions, and into nothing at all once the weight has
// reached max_weight.
__global__ __launch_bounds__(64) void det_apply_long_kernel(TsdfLayerDev L, vgx_tsdf_config c, size_t M,
                                                           const uint32_t* __restrict__ c_key,
                                                           const long long* __restrict__ t_at,
                                                           const float* __restrict__ t_sdf,
                                                           const float* __restrict__ t_w,
                                                           const uint32_t* __restrict__ t_color,
                                                           const uint8_t* __restrict__ t_far,
                                                           const uint32_t* __restrict__ long_runs,
                                                           unsigned long long* __restrict__ ctr) {
  constexpr int E = 8;  // sub-batches of 64 in flight: the chain of a batch is short, the loads are not
  const int lane = threadIdx.x;
  const unsigned long long n_long = ctr[kCtrLong];
  const float trunc = c.default_truncation_distance;
  unsigned long long dropped = 0;
  for (unsigned long long j = blockIdx.x; j < n_long; j += gridDim.x) {
    const size_t q0 = long_runs[j];
    const uint32_t key = c_key[q0];
    long long at_cached = -1;
    float d = 0.0f, w = 0.0f;
    uint32_t col = 0u;
    bool dirty = false;
    bool n_valid[E], n_far[E];
    long long n_at[E];
    float n_sdf[E], n_uw[E];
    uint32_t n_color[E];
#pragma unroll
    for (int e = 0; e < E; ++e) {
      // (all loads unconditional on a clamped index: none waits for the key comparison)
      const size_t qn = q0 + (size_t)e * 64 + lane, qc = qn < M ? qn : M - 1;
      const uint32_t k_ = c_key[qc];
      n_at[e] = t_at[qc];
      n_sdf[e] = t_sdf[qc];
      n_uw[e] = t_w[qc];
      n_color[e] = t_color[qc];
      n_far[e] = t_far[qc] != 0;
      n_valid[e] = qn < M && k_ == key;
    }
    bool run_ended = false;
    for (size_t b = q0; !run_ended; b += (size_t)E * 64) {
      bool valid_[E], far_[E];
      long long at_[E];
      float sdf_[E], uw_[E];
      uint32_t color_[E];
#pragma unroll
      for (int e = 0; e < E; ++e) {
        valid_[e] = n_valid[e]; at_[e] = n_at[e]; sdf_[e] = n_sdf[e]; uw_[e] = n_uw[e]; color_[e] = n_color[e];
        far_[e] = n_far[e];
      }
      // the valid lanes are a prefix of the E * 64 (the run is contiguous): a full last sub-batch means the
      // run may go on -- fetch the next E sub-batches while these are applied
      if (__popcll(__ballot(valid_[E - 1])) == 64) {
#pragma unroll
        for (int e = 0; e < E; ++e) {
          const size_t qn = b + (size_t)(E + e) * 64 + lane, qc = qn < M ? qn : M - 1;
          const uint32_t k_ = c_key[qc];
          n_at[e] = t_at[qc];
          n_sdf[e] = t_sdf[qc];
          n_uw[e] = t_w[qc];
          n_color[e] = t_color[qc];
          n_far[e] = t_far[qc] != 0;
          n_valid[e] = qn < M && k_ == key;
        }
      } else {
        run_ended = true;
      }
#pragma unroll
      for (int e = 0; e < E; ++e) {
        const bool valid = valid_[e];
        const long long at = valid ? at_[e] : -1;
        const float sdf = sdf_[e], uw = uw_[e];
        const uint32_t color = color_[e];
        const int cnt = __popcll(__ballot(valid));
        if (cnt == 0) break;
        const long long at0 = ((long long)__builtin_amdgcn_readfirstlane((int)(at >> 32)) << 32) |
                              (unsigned int)__builtin_amdgcn_readfirstlane((int)(at & 0xffffffffll));
        const bool one_voxel = at0 >= 0 && __ballot(valid && at != at0) == 0ull;
        bool done = false;
        if (one_voxel) {
          if (at0 != at_cached) {
            if (dirty && lane == 0) {
              L.voxels[at_cached] = pack_voxel(d, w);
              L.rgba[at_cached] = col;
            }
            const unsigned long long v = L.voxels[at0];
            d = __uint_as_float((unsigned)(v & 0xffffffffull));
            w = __uint_as_float((unsigned)(v >> 32));
            col = L.rgba[at0];
            at_cached = at0;
            dirty = false;
          }
          // every update of the sub-batch is "far" (sdf >= 2 t, 1e5 w >= max_weight >= W) and the voxel sits at
          // +t with W <= max_weight: every step only adds its weight (see above) -- one ballot decides
          if (d == trunc && w <= c.max_weight && __ballot(valid && !far_[e]) == 0ull) {  // (far_ of invalid lanes: ignored)
            if (w != c.max_weight) {
              for (int k = 0; k < cnt; ++k)
                w = fminf(c.max_weight, w + __int_as_float(__builtin_amdgcn_readlane(__float_as_int(uw), k)));
              dirty = true;
            }
            done = true;  // (at max_weight already: min(max_weight, max_weight + uw) = max_weight, nothing moves)
          }
        }
        if (!done) {
          for (int k = 0; k < cnt; ++k) {
            const long long at_k = ((long long)__builtin_amdgcn_readlane((int)(at >> 32), k) << 32) |
                                   (unsigned int)__builtin_amdgcn_readlane((int)(at & 0xffffffffll), k);
            if (at_k < 0) {
              ++dropped;
              continue;
            }
            if (at_k != at_cached) {
              if (dirty && lane == 0) {
                L.voxels[at_cached] = pack_voxel(d, w);
                L.rgba[at_cached] = col;
              }
              const unsigned long long v = L.voxels[at_k];
              d = __uint_as_float((unsigned)(v & 0xffffffffull));
              w = __uint_as_float((unsigned)(v >> 32));
              col = L.rgba[at_k];
              at_cached = at_k;
              dirty = false;
            }
            const float sdf_k = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(sdf), k));
            const float uw_k = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(uw), k));
            if (d == trunc && sdf_k >= 2.0f * trunc && uw_k * 1.0e5f >= w && w + uw_k >= 1e-6f) {
              w = fminf(c.max_weight, w + uw_k);  // the average stays at +truncation: weight only
              dirty = true;
            } else {
              const uint32_t color_k = (uint32_t)__builtin_amdgcn_readlane((int)color, k);
              apply_term(c, sdf_k, uw_k, color_k, d, w, col, dirty);
            }
          }
        }
      }
    }
    if (dirty && lane == 0) {
      L.voxels[at_cached] = pack_voxel(d, w);
      L.rgba[at_cached] = col;
    }
  }
  if (dropped && lane == 0) {
    atomicAdd(L.dropped, dropped);
    atomicAdd(&ctr[kCtrDropped], dropped);
  }
}

// out[i] = in[0] + ... + in[i - 1], i < n, in one launch (TileChain; tiles of 1024, four consecutive items per thread)
constexpr int kScanIpt = 4;
__global__ __launch_bounds__(256) void det_offsets_kernel(const uint32_t* __restrict__ in, uint32_t* __restrict__ out, uint32_t n,
                                                         TileChain chain) {
  __shared__ uint32_t sh_word, sh4[4];
  const uint32_t tile = chain_tile(chain, &sh_word);
  const size_t base = ((size_t)tile * 256 + threadIdx.x) * kScanIpt;
  uint32_t v[kScanIpt], mine = 0;
#pragma unroll
  for (int e = 0; e < kScanIpt; ++e) {
    v[e] = base + e < n ? in[base + e] : 0u;
    mine += v[e];
  }
  uint32_t in_tile = 0;
  const uint32_t before = block_exclusive_sum(mine, sh4, in_tile);
  uint32_t run = chain_exclusive_sum(chain, tile, in_tile, &sh_word) + before;
#pragma unroll
  for (int e = 0; e < kScanIpt; ++e) {
    if (base + e < n) out[base + e] = run;
    run += v[e];
  }
}

__global__ void det_fill_u64_kernel(unsigned long long* p, size_t n, unsigned long long v) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) p[i] = v;
}

// ---- merged integrator: every group's ray written out (cast from the origin outwards) ---------------
constexpr unsigned long long kMergedKeyBias = 1ull << 20;
__device__ __forceinline__ unsigned long long merged_voxel_key(int x, int y, int z) {
  return (((unsigned long long)(x + (long long)kMergedKeyBias) & 0x1fffffull) << 42) |
         (((unsigned long long)(y + (long long)kMergedKeyBias) & 0x1fffffull) << 21) |
         ((unsigned long long)(z + (long long)kMergedKeyBias) & 0x1fffffull);
}

// 16 lanes per group: the DDA is a sequential f32 recurrence, so all 16 run it in lock step (the same
// arithmetic as the single-thread walk, hence the same voxels) and lane l keeps step base + l: the records
// of 16 consecutive steps are written side by side instead of one thread striding through its ray.
__global__ __launch_bounds__(256) void det_merged_walk_kernel(vgx_tsdf_config c, float vsi, float tx, float ty, float tz,
                                                             uint32_t G, const float4* __restrict__ g_pg,
                                                             const uint32_t* __restrict__ g_flags,
                                                             const uint32_t* __restrict__ g_count,
                                                             const uint32_t* __restrict__ off,
                                                             const unsigned long long* __restrict__ keys,
                                                             const unsigned int* __restrict__ group_start,
                                                             const unsigned int* __restrict__ counters, int anti_grazing,
                                                             unsigned long long* __restrict__ acc_vox,
                                                             uint32_t* __restrict__ acc_key, uint32_t* __restrict__ acc_ray,
                                                             uint8_t* __restrict__ upd,
                                                             unsigned long long* __restrict__ ctr) {
  constexpr int LANES = 16;
  const int lane = threadIdx.x & (LANES - 1);
  const uint32_t n_groups = min(G, counters[0]);
  const long long n_surface = counters[1];
  const uint32_t n_sub = gridDim.x * (blockDim.x / LANES);
  bool out_of_range = false;
  for (uint32_t g = blockIdx.x * (blockDim.x / LANES) + threadIdx.x / LANES; g < n_groups; g += n_sub) {
    const uint32_t cnt = g_count[g];
    if (cnt == 0) continue;
    const bool clearing_ray = (g_flags[g] & 2u) != 0;
    const float4 pg = g_pg[g];
    const unsigned long long own_key = keys[group_start[g]] & ~(1ull << 63);
    RayDda r = ray_setup(c, vsi, tx, ty, tz, pg.x, pg.y, pg.z, clearing_ray, true);
    const size_t base = off[g];
    for (uint32_t k0 = 0; k0 < cnt; k0 += LANES) {
      int vx = 0, vy = 0, vz = 0;
#pragma unroll
      for (int i = 0; i < LANES; ++i) {
        if (i == lane) {
          vx = r.curr[0]; vy = r.curr[1]; vz = r.curr[2];
        }
        dda_advance(r);
      }
      const uint32_t k = k0 + lane;
      if (k >= cnt) continue;
      out_of_range |= vx <= -kVoxBias || vx >= kVoxBias || vy <= -kVoxBias || vy >= kVoxBias || vz <= -kVoxBias || vz >= kVoxBias;
      uint8_t u = 1;
      if (anti_grazing) {  // skip voxels that are the end voxel of another surface group (voxel_map.find)
        const unsigned long long key = merged_voxel_key(vx, vy, vz);
        if (clearing_ray || key != own_key) {
          long long lo = 0, hi = n_surface;
          while (lo < hi) {
            const long long mid = (lo + hi) >> 1;
            if (keys[mid] < key) lo = mid + 1; else hi = mid;
          }
          if (lo < n_surface && keys[lo] == key) u = 0;
        }
      }
      acc_vox[base + k] = pack_vox(vx, vy, vz);
      acc_key[base + k] = index_hash(vx, vy, vz) & kSetMask;
      acc_ray[base + k] = g;
      upd[base + k] = u;
    }
  }
  if (out_of_range) ctr[kCtrError] = 2ull;
}

// ---------------------------------------------------------------------------------------------------
int grow(vgx_ctx ctx, Buf& b, size_t bytes) {
  if (b.bytes >= bytes) return VGX_OK;
  VGX_HIP(ctx, hipStreamSynchronize(ctx->tsdf_stream));
  if (b.p) (void)hipFree(b.p);
  b.p = nullptr;
  b.bytes = 0;
  const size_t want = std::max<size_t>(bytes + bytes / 4, 256);  // a quarter of slack: scans of a session vary
  if (hipMalloc(&b.p, want) != hipSuccess) {
    (void)hipGetLastError();
    return set_error(ctx, VGX_ERR_NOMEM, "TSDF reproducible mode: scratch allocation failed (" + std::to_string(want) + " bytes)");
  }
  b.bytes = want;
  return VGX_OK;
}

inline unsigned blocks_for(size_t n) { return (unsigned)((n + 255) / 256); }

#define DET_TRY(expr)             \
  do {                            \
    int rc_ = (expr);             \
    if (rc_ != VGX_OK) return rc_; \
  } while (0)

// experiment aid (VGX_DET_PHASES=1): host-side time between the points of a scan where the host has waited for
// the stream, so what is queued in between is charged to the wait after it
struct PhaseClock {
  bool on;
  std::chrono::steady_clock::time_point t;
  double us[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  long long sweeps = 0, attempts = 0;
  PhaseClock() {
    static const bool enabled = getenv("VGX_DET_PHASES") != nullptr;
    on = enabled;
    if (on) t = std::chrono::steady_clock::now();
  }
  void mark(int phase) {
    if (!on) return;
    const auto now = std::chrono::steady_clock::now();
    us[phase] += std::chrono::duration<double, std::micro>(now - t).count();
    t = now;
  }
  void print(size_t N) const {
    if (!on) return;
    fprintf(stderr, "[vgx det phases] to count read-back %.0f us | walk+sort+gather+first sweep %.0f | other sweeps %.0f | "
                    "extend %.0f | commit to its read-back %.0f | rest queued %.0f   (%lld attempts, %lld sweeps, %zu accesses)\n",
            us[0], us[1], us[2], us[3], us[4], us[5], attempts, sweeps, N);
  }
};

// a 4-byte device value into one of the pinned words behind the counters' mirror (valid after read_counters)
int fetch_u32(vgx_ctx ctx, DetScratch* S, int word, const void* dev) {
  S->h_ctr[word] = 0ull;
  VGX_HIP(ctx, hipMemcpyAsync(&S->h_ctr[word], dev, 4, hipMemcpyDeviceToHost, ctx->tsdf_stream));
  return VGX_OK;
}

int grow_zeroed(vgx_ctx ctx, Buf& b, size_t bytes) {
  if (b.bytes >= bytes) return VGX_OK;
  int rc = grow(ctx, b, bytes);
  if (rc != VGX_OK) return rc;
  VGX_HIP(ctx, hipMemsetAsync(b.p, 0, b.bytes, ctx->tsdf_stream));
  return VGX_OK;
}

// Until the sweep numbered `seq`, or a later one, has reported through the pinned word.  No stream call on the
// way as long as the report arrives; a stream that has gone idle without it (or has failed) ends the wait.
int wait_for_sweep(vgx_ctx ctx, DetScratch* S, unsigned long long seq, unsigned long long* word) {
  const auto t0 = std::chrono::steady_clock::now();
  for (unsigned spins = 1;; ++spins) {
    const unsigned long long w = __atomic_load_n(S->h_flag, __ATOMIC_ACQUIRE);
    if ((w >> 2) >= seq) {
      *word = w;
      return VGX_OK;
    }
    if (spins % 8192u != 0) {
      __builtin_ia32_pause();
      continue;
    }
    const hipError_t e = hipStreamQuery(ctx->tsdf_stream);
    if (e == hipSuccess) {  // everything queued has run: the report is there, or it never will be
      const unsigned long long w2 = __atomic_load_n(S->h_flag, __ATOMIC_ACQUIRE);
      if ((w2 >> 2) >= seq) {
        *word = w2;
        return VGX_OK;
      }
      return set_error(ctx, VGX_ERR_HIP, "TSDF reproducible mode: a sweep ended without reporting (internal error)");
    }
    if (e != hipErrorNotReady) VGX_HIP(ctx, e);
    if (std::chrono::steady_clock::now() - t0 > std::chrono::seconds(30))
      return set_error(ctx, VGX_ERR_HIP, "TSDF reproducible mode: no report from a sweep within 30 s");
  }
}

// keys (end_bit bits) -> keys_sorted, and where every sorted key came from; stable
int sort_by_slot(vgx_ctx ctx, DetScratch* S, const uint32_t* keys, uint32_t* keys_sorted, uint32_t* idx_sorted, size_t n,
                 unsigned end_bit) {
  size_t bytes = 0;
  auto iota = rocprim::make_counting_iterator<uint32_t>(0u);
  VGX_HIP(ctx, stable_sort_pairs(nullptr, bytes, keys, keys_sorted, iota, idx_sorted, n, end_bit, ctx->tsdf_stream));
  int rc = grow(ctx, S->tmp, bytes);
  if (rc != VGX_OK) return rc;
  bytes = S->tmp.bytes;
  VGX_HIP(ctx, stable_sort_pairs(S->tmp.p, bytes, keys, keys_sorted, iota, idx_sorted, n, end_bit, ctx->tsdf_stream));
  return VGX_OK;
}

// the chain of the next prefix-sum launch over `tiles` workgroups: room for its words, its epoch, its ticket base
int next_chain(vgx_ctx ctx, DetScratch* S, uint32_t tiles, TileChain* ch) {
  int rc = grow_zeroed(ctx, S->tile_state, (size_t)tiles * 8);
  if (rc != VGX_OK) return rc;
  if (S->sweep_epoch >= kChainEpochMax) {  // (after 2^30 launches: start the tags over)
    VGX_HIP(ctx, hipMemsetAsync(S->tile_state.p, 0, S->tile_state.bytes, ctx->tsdf_stream));
    S->sweep_epoch = 0;
  }
  ++S->sweep_epoch;
  ch->state = S->tile_state.as<unsigned long long>();
  ch->epoch = S->sweep_epoch;
  ch->ticket = S->d_ctr + kCtrChainTicket;
  ch->ticket_base = S->chain_tickets;
  ch->error = S->d_ctr + kCtrError;
  S->chain_tickets += tiles;
  return VGX_OK;
}

constexpr int kReportWords = 32;  // [0] sequence number, [2 .. 2 + kCtrCount) the counters
static_assert(2 + kCtrCount <= kReportWords, "the report holds every counter");

__global__ __launch_bounds__(64) void det_report_kernel(const unsigned long long* __restrict__ ctr,
                                                       unsigned long long* __restrict__ host_words, unsigned long long seq) {
  const int i = (int)threadIdx.x;
  if (i < kCtrCount)
    __hip_atomic_store(&host_words[2 + i], __hip_atomic_load(&ctr[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT), __ATOMIC_RELAXED,
                       __HIP_MEMORY_SCOPE_SYSTEM);
  __threadfence_system();
  __syncthreads();
  if (i == 0) __hip_atomic_store(&host_words[0], seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}

// The scan's counters on the host, and everything queued before this call finished.  Through the pinned report words:
// one tiny kernel behind the producers and a poll (a few microseconds after the producer ends) instead of a copy through
// the blit path and a stream synchronisation (20-45 us of bubble per read-back in profiles/r05_tsdf_launches.txt, two per
// scan).  VGX_DET_REPORT=0: the copy + synchronise of rounds 3-4 (A/B aid; also the fallback without pinned memory).
int read_counters(vgx_ctx ctx, DetScratch* S) {
  static const bool by_report = !(getenv("VGX_DET_REPORT") && atoi(getenv("VGX_DET_REPORT")) == 0);
  if (!by_report || !S->h_report) {
    VGX_HIP(ctx, hipMemcpyAsync(S->h_ctr, S->d_ctr, kCtrCount * 8, hipMemcpyDeviceToHost, ctx->tsdf_stream));
    VGX_HIP(ctx, hipStreamSynchronize(ctx->tsdf_stream));
    return VGX_OK;
  }
  const unsigned long long seq = ++S->report_seq;
  hipLaunchKernelGGL(det_report_kernel, dim3(1), dim3(64), 0, ctx->tsdf_stream, S->d_ctr, S->d_report, seq);
  VGX_HIP(ctx, hipGetLastError());
  const auto t0 = std::chrono::steady_clock::now();
  for (unsigned spins = 1;; ++spins) {
    if (__atomic_load_n(&S->h_report[0], __ATOMIC_ACQUIRE) >= seq) break;
    if (spins % 8192u != 0) {
      __builtin_ia32_pause();
      continue;
    }
    const hipError_t e = hipStreamQuery(ctx->tsdf_stream);
    if (e == hipSuccess) {  // everything queued has run: the report is there, or it never will be
      if (__atomic_load_n(&S->h_report[0], __ATOMIC_ACQUIRE) >= seq) break;
      return set_error(ctx, VGX_ERR_HIP, "TSDF reproducible mode: the counters' report never arrived (internal error)");
    }
    if (e != hipErrorNotReady) VGX_HIP(ctx, e);
    if (std::chrono::steady_clock::now() - t0 > std::chrono::seconds(30))
      return set_error(ctx, VGX_ERR_HIP, "TSDF reproducible mode: no report of the counters within 30 s");
  }
  for (int i = 0; i < kCtrCount; ++i) S->h_ctr[i] = __atomic_load_n(&S->h_report[2 + i], __ATOMIC_RELAXED);
  return VGX_OK;
}

}  // namespace

static int ensure_scratch(vgx_tsdf_integrator I) {
  vgx_ctx ctx = I->ctx;
  if (I->det) return VGX_OK;
  I->det = new (std::nothrow) DetScratch();
  if (!I->det) return set_error(ctx, VGX_ERR_NOMEM, "TSDF reproducible mode: out of host memory");
  DetScratch* S = I->det;
  if (hipMalloc(&S->d_ctr, kCtrCount * 8) != hipSuccess ||
      hipHostMalloc((void**)&S->h_ctr, kHostWords * 8, hipHostMallocDefault) != hipSuccess)
    return set_error(ctx, VGX_ERR_NOMEM, "TSDF reproducible mode: counter allocation failed");
  // the word the sweeps report through: host memory the kernel writes while it runs and the host reads while
  // the stream is busy, so it has to be coherent (fine-grained), not merely pinned
  if (hipHostMalloc((void**)&S->h_flag, 64, hipHostMallocMapped | hipHostMallocCoherent) != hipSuccess) {
    (void)hipGetLastError();
    S->h_flag = nullptr;
    if (hipHostMalloc((void**)&S->h_flag, 64, hipHostMallocMapped) != hipSuccess)
      return set_error(ctx, VGX_ERR_NOMEM, "TSDF reproducible mode: flag allocation failed");
  }
  S->h_flag[0] = 0ull;
  void* dp = nullptr;
  if (hipHostGetDevicePointer(&dp, S->h_flag, 0) != hipSuccess || !dp) {
    (void)hipGetLastError();
    dp = S->h_flag;  // (unified addressing: the same pointer)
  }
  S->d_flag = (unsigned long long*)dp;
  // the counters' report words: coherent or nothing (read_counters then copies and synchronises as before)
  if (hipHostMalloc((void**)&S->h_report, kReportWords * 8, hipHostMallocMapped | hipHostMallocCoherent) != hipSuccess) {
    (void)hipGetLastError();
    S->h_report = nullptr;
  } else {
    for (int i = 0; i < kReportWords; ++i) S->h_report[i] = 0ull;
    void* rp = nullptr;
    if (hipHostGetDevicePointer(&rp, S->h_report, 0) != hipSuccess || !rp) {
      (void)hipGetLastError();
      rp = S->h_report;
    }
    S->d_report = (unsigned long long*)rp;
  }
  return VGX_OK;
}

// the scan's counters where vgx_tsdf.hip's merged-integrator kernels write them (the scratch object is made on demand)
int det_counters(vgx_tsdf_integrator I, unsigned long long** d_ctr) {
  DET_TRY(ensure_scratch(I));
  *d_ctr = I->det->d_ctr;
  I->det->chain_tickets = 0;  // (the caller's first kernel zeroes the counters)
  return VGX_OK;
}

// the TileChain of the merged integrator's next prefix-sum launch (vgx_tsdf.hip), after det_counters
int det_next_chain(vgx_tsdf_integrator I, uint32_t tiles, TileChain* chain) {
  return next_chain(I->ctx, I->det, tiles, chain);
}

// ---- 3. the updates that happen, in sorted order -> compaction, new blocks, ordered application ----
// (shared by both integrators: `update` says which of the N sorted accesses update their voxel; rays are
// indexed by acc_ray, their point / weight / colour live in ray_pg / ray_color)
// what a non-zero kCtrError means (vgx_tsdf_internal.h: 1-3 from the reproducible mode's kernels and the chained
// prefix sums, 4-5 from the merged integrator's)
static int det_error_code(vgx_ctx ctx, unsigned long long code) {
  switch (code) {
    case 0ull: return VGX_OK;
    case 1ull: return set_error(ctx, VGX_ERR_UNSUPPORTED, "TSDF reproducible mode: a ray longer than 2^24 voxel steps");
    case 2ull: return set_error(ctx, VGX_ERR_UNSUPPORTED, "TSDF reproducible mode: a voxel index beyond +-2^20 voxels of the layer origin");
    case 3ull: return set_error(ctx, VGX_ERR_HIP, "TSDF reproducible mode: a tile of a chained kernel never reported (internal error)");
    case kErrMergedRayTooLong: return set_error(ctx, VGX_ERR_UNSUPPORTED, "TSDF merged integrator: a ray longer than 2^24 voxel steps");
    case kErrMergedKeyCorner:
      return set_error(ctx, VGX_ERR_UNSUPPORTED,
                       "TSDF merged integrator: a clearing point whose end voxel is 2^20 - 1 (mod 2^21) on all three axes cannot be keyed");
    default: return set_error(ctx, VGX_ERR_HIP, "TSDF reproducible mode: unknown device error code " + std::to_string(code));
  }
}

static int det_commit(vgx_tsdf_integrator I, DetScratch* S, const float T[7], size_t N,
                      const UpdateOp* update, bool update_set, bool ordered_blocks, const float4* ray_pg,
                      const uint32_t* ray_color, int64_t* n_updates, PhaseClock* pc = nullptr) {
  vgx_ctx ctx = I->ctx;
  vgx_tsdf_layer layer = I->layer;
  const vgx_tsdf_config& c = I->dev.cfg;
  hipStream_t st = ctx->tsdf_stream;
  size_t M = N;
  const uint32_t* c_idx = S->s_idx.as<uint32_t>();  // without a filter every sorted access is an update
  const uint32_t* c_key = S->s_key.as<uint32_t>();
  const uint32_t* M_dev = nullptr;  // where the device keeps the number of updates (cpos[N]) until it has been read back
  if (update) {
    // `last` is consumed by the finish kernel (set state); the number of updates lands behind the sort's input keys
    uint32_t* m_word = S->acc_key.as<uint32_t>() + N;  // (free since the sort)
    const uint32_t finish_tiles = (uint32_t)((N + kSweepTile - 1) / kSweepTile);
    TileChain chain;
    DET_TRY(next_chain(ctx, S, finish_tiles, &chain));
    if (update->flags)
      hipLaunchKernelGGL(det_finish_kernel<true>, dim3(finish_tiles), dim3(256), 0, st, (uint32_t)N, S->s_key.as<uint32_t>(),
                         S->s_idx.as<uint32_t>(), S->s_h.as<uint32_t>(), S->last.as<uint32_t>(), *update, chain, m_word,
                         S->c_idx.as<uint32_t>(), S->c_key.as<uint32_t>(), update_set ? I->dev.observed_set : nullptr,
                         I->dev.observed_offset);
    else
      hipLaunchKernelGGL(det_finish_kernel<false>, dim3(finish_tiles), dim3(256), 0, st, (uint32_t)N, S->s_key.as<uint32_t>(),
                         S->s_idx.as<uint32_t>(), S->s_h.as<uint32_t>(), S->last.as<uint32_t>(), *update, chain, m_word,
                         S->c_idx.as<uint32_t>(), S->c_key.as<uint32_t>(), update_set ? I->dev.observed_set : nullptr,
                         I->dev.observed_offset);
    VGX_HIP(ctx, hipGetLastError());
    M_dev = m_word;
    c_idx = S->c_idx.as<uint32_t>();
    c_key = S->c_key.as<uint32_t>();
  }
  TsdfLayerDev& L = layer->dev;
  // ONE read-back for the whole commit (round 3: three, each a stream round trip of 20-40 us): the number of
  // updates M, the new blocks they need, the layer's block count.  What has to run before it is launched over
  // the upper bound N and looks M up on the device.
  size_t new_cap = 0;
  if (ordered_blocks) {
    // new blocks take their pool slots in the order of their first update (the table may have been
    // re-boxed since the last scan)
    if (S->first_touch.bytes < layer->lut_cells * 8 || S->first_touch_dirty) {  // else: all ~0 since the last scan
      DET_TRY(grow(ctx, S->first_touch, layer->lut_cells * 8));
      S->first_touch_dirty = false;
      const size_t cells = S->first_touch.bytes / 8;
      hipLaunchKernelGGL(det_fill_u64_kernel, dim3(blocks_for(cells)), dim3(256), 0, st, S->first_touch.as<unsigned long long>(),
                         cells, ~0ull);
      VGX_HIP(ctx, hipGetLastError());
    }
    new_cap = (size_t)std::max<int64_t>(tsdf_last_scan_bound(layer), 1);
    DET_TRY(grow(ctx, S->new_cells, new_cap * 4));
    DET_TRY(grow(ctx, S->new_cells_sorted, new_cap * 4));
    DET_TRY(grow(ctx, S->new_keys, new_cap * 8));
    DET_TRY(grow(ctx, S->new_keys_sorted, new_cap * 8));
    hipLaunchKernelGGL(det_blocks_kernel, dim3(blocks_for(N)), dim3(256), 0, st, L, N, M_dev, c_idx,
                       S->acc_vox.as<unsigned long long>(), S->acc_ray.as<uint32_t>(), S->off.as<uint32_t>(),
                       S->first_touch.as<unsigned long long>(), S->new_cells.as<int32_t>(), (uint32_t)new_cap, S->d_ctr);
    VGX_HIP(ctx, hipGetLastError());
    // from here until det_assign_kernel has put the marked cells back to ~0, a failure leaves marks behind:
    // the next scan refills the table
    S->first_touch_dirty = true;
  }
  int32_t n_blocks_now = 0;
  if (M_dev && !ordered_blocks) DET_TRY(fetch_u32(ctx, S, kHostM, M_dev));  // (else: det_blocks_kernel left it in the counters)
  if (M_dev || ordered_blocks) {
    DET_TRY(read_counters(ctx, S));
    // a look-back that timed out substituted a zero prefix (3), a walk left the +-2^20 band (2): the compaction
    // would be wrong -- nothing is applied to the layer
    if (S->h_ctr[kCtrError]) return det_error_code(ctx, S->h_ctr[kCtrError]);
  }
  if (pc) pc->mark(4);
  if (ordered_blocks) {
    M = (size_t)S->h_ctr[kCtrM];
    n_blocks_now = (int32_t)(uint32_t)S->h_ctr[kCtrBlocks];
  } else if (M_dev) {
    M = (uint32_t)S->h_ctr[kHostM];
  }
  if (M == 0) {
    S->first_touch_dirty = false;  // nothing was marked
    return VGX_OK;
  }
  if (ordered_blocks) {
    const size_t n_new = (size_t)S->h_ctr[kCtrNew];
    if (n_new > new_cap)
      return set_error(ctx, VGX_ERR_HIP, "TSDF reproducible mode: more new blocks than the scan's reach allows (internal error)");
    if (n_new > 0) {
      hipLaunchKernelGGL(det_new_keys_kernel, dim3(blocks_for(n_new)), dim3(256), 0, st, (uint32_t)n_new,
                         S->new_cells.as<int32_t>(), S->first_touch.as<unsigned long long>(),
                         S->new_keys.as<unsigned long long>());
      VGX_HIP(ctx, hipGetLastError());
      size_t bytes = 0;
      VGX_HIP(ctx, rocprim::radix_sort_pairs(nullptr, bytes, S->new_keys.as<unsigned long long>(),
                                             S->new_keys_sorted.as<unsigned long long>(), S->new_cells.as<int32_t>(),
                                             S->new_cells_sorted.as<int32_t>(), n_new, 0, 64, st));
      DET_TRY(grow(ctx, S->tmp, bytes));
      bytes = S->tmp.bytes;
      VGX_HIP(ctx, rocprim::radix_sort_pairs(S->tmp.p, bytes, S->new_keys.as<unsigned long long>(),
                                             S->new_keys_sorted.as<unsigned long long>(), S->new_cells.as<int32_t>(),
                                             S->new_cells_sorted.as<int32_t>(), n_new, 0, 64, st));
      hipLaunchKernelGGL(det_assign_kernel, dim3(blocks_for(n_new)), dim3(256), 0, st, L, (uint32_t)n_new, n_blocks_now,
                         S->new_cells_sorted.as<int32_t>(), S->first_touch.as<unsigned long long>());
      VGX_HIP(ctx, hipGetLastError());
    }
    S->first_touch_dirty = false;  // (n_new == 0: nothing was marked)
  }
  DET_TRY(grow(ctx, S->long_runs, (M / kShortRun + 2) * 4));
  DET_TRY(grow(ctx, S->t_at, M * 8));
  DET_TRY(grow(ctx, S->t_sdf, M * 4));
  DET_TRY(grow(ctx, S->t_w, M * 4));
  DET_TRY(grow(ctx, S->t_color, M * 4));
  DET_TRY(grow(ctx, S->t_far, M));
  if (ordered_blocks)
    hipLaunchKernelGGL(det_terms_kernel<false>, dim3(blocks_for(M)), dim3(256), 0, st, L, c, T[4], T[5], T[6], M, c_idx,
                       S->acc_vox.as<unsigned long long>(), S->acc_ray.as<uint32_t>(), ray_pg, ray_color,
                       S->t_at.as<long long>(), S->t_sdf.as<float>(), S->t_w.as<float>(), S->t_color.as<uint32_t>(),
                       S->t_far.as<uint8_t>());
  else
    hipLaunchKernelGGL(det_terms_kernel<true>, dim3(blocks_for(M)), dim3(256), 0, st, L, c, T[4], T[5], T[6], M, c_idx,
                       S->acc_vox.as<unsigned long long>(), S->acc_ray.as<uint32_t>(), ray_pg, ray_color,
                       S->t_at.as<long long>(), S->t_sdf.as<float>(), S->t_w.as<float>(), S->t_color.as<uint32_t>(),
                       S->t_far.as<uint8_t>());
  VGX_HIP(ctx, hipGetLastError());
  hipLaunchKernelGGL(det_apply_kernel, dim3(blocks_for(M)), dim3(256), 0, st, L, c, M, c_key, S->t_at.as<long long>(),
                     S->t_sdf.as<float>(), S->t_w.as<float>(), S->t_color.as<uint32_t>(), S->long_runs.as<uint32_t>(), S->d_ctr);
  const unsigned long_grid = (unsigned)std::min<size_t>((size_t)ctx->cu_count * 32, M / kShortRun + 1);
  hipLaunchKernelGGL(det_apply_long_kernel, dim3(long_grid), dim3(64), 0, st, L, c, M, c_key, S->t_at.as<long long>(),
                     S->t_sdf.as<float>(), S->t_w.as<float>(), S->t_color.as<uint32_t>(), S->t_far.as<uint8_t>(),
                     S->long_runs.as<uint32_t>(), S->d_ctr);
  VGX_HIP(ctx, hipGetLastError());
  static const bool debug = getenv("VGX_DET_DEBUG") != nullptr;  // experiment aid: sizes of the ordered application
  if (debug) {
    DET_TRY(read_counters(ctx, S));
    fprintf(stderr, "[vgx det] updates %zu, long runs %llu (of more than %d), accesses sorted %zu\n", M,
            (unsigned long long)S->h_ctr[kCtrLong], kShortRun, N);
  }
  if (n_updates) {
    DET_TRY(read_counters(ctx, S));
    *n_updates = (int64_t)M - (int64_t)S->h_ctr[kCtrDropped];
  }
  if (pc) {
    pc->mark(5);
    pc->print(N);
  }
  return VGX_OK;
}

int det_integrate(vgx_tsdf_integrator I, const float T[7], const void* d_points, const void* d_rgba, int64_t n,
                  int32_t freespace, const uint32_t* order, int64_t* n_updates) {
  vgx_ctx ctx = I->ctx;
  vgx_tsdf_layer layer = I->layer;
  const vgx_tsdf_config& c = I->dev.cfg;
  hipStream_t st = ctx->tsdf_stream;
  if (n_updates) *n_updates = 0;
  if (n >= (1ll << 31)) return set_error(ctx, VGX_ERR_UNSUPPORTED, "TSDF reproducible mode: more than 2^31 points in a scan");
  DET_TRY(ensure_scratch(I));
  DetScratch* S = I->det;
  PhaseClock pc;
  const float vsi = layer->dev.voxel_size_inv;
  const size_t np = (size_t)n;
  // ---- 1. points, start set ----
  DET_TRY(grow(ctx, S->ray_pg, np * 16));
  DET_TRY(grow(ctx, S->ray_color, np * 4));
  DET_TRY(grow(ctx, S->ray_flags, np * 4));
  DET_TRY(grow(ctx, S->start_val, np * 8));
  DET_TRY(grow(ctx, S->start_key, np * 4));
  DET_TRY(grow(ctx, S->start_key_sorted, np * 4));
  DET_TRY(grow(ctx, S->start_seq_sorted, np * 4));
  DET_TRY(grow(ctx, S->count, (np + 1) * 4));
  DET_TRY(grow(ctx, S->off, (np + 1) * 4));
  DET_TRY(grow(ctx, S->T, np * 4));
  DET_TRY(grow(ctx, S->broke, np));
  DET_TRY(grow(ctx, S->full_count, np * 4));
  DET_TRY(grow(ctx, S->ext, np + 1));
  DET_TRY(grow(ctx, S->walked, np * 4));
  // (a scan whose complete walks were too many to write out is usually followed by another one: a depth camera
  // stays a depth camera.  Then the walks are cut at once and the count of the complete ones is skipped; the
  // result does not depend on where rays are cut, det_count_kernel.)
  const bool may_cap_at_all = c.max_consecutive_ray_collisions < (1 << 20);
  const bool start_capped = may_cap_at_all && S->start_capped;
  const bool keep_marks = start_capped && S->ext_points == n;  // the previous scan's pioneers: written out completely at once
  S->ext_points = -1;  // (until this scan has left its own)
  S->chain_tickets = 0;  // (det_points_kernel zeroes the counters)
  hipLaunchKernelGGL(det_points_kernel, dim3(blocks_for(np + 1)), dim3(256), 0, st, c, vsi, T[0], T[1], T[2], T[3], T[4], T[5],
                     T[6], (const float*)d_points, (const uint32_t*)d_rgba, (long long)n, order, (int)freespace,
                     I->dev.start_offset, S->ray_pg.as<float4>(), S->ray_color.as<uint32_t>(), S->ray_flags.as<uint32_t>(),
                     S->start_val.as<unsigned long long>(), S->start_key.as<uint32_t>(), S->ext.as<uint8_t>(),
                     keep_marks ? 2 : start_capped ? 0 : 1, S->d_ctr);
  VGX_HIP(ctx, hipGetLastError());
  // stable: equal slots keep the visiting order they were written in
  DET_TRY(sort_by_slot(ctx, S, S->start_key.as<uint32_t>(), S->start_key_sorted.as<uint32_t>(), S->start_seq_sorted.as<uint32_t>(),
                       np, kSetBits + 1));
  hipLaunchKernelGGL(det_start_kernel, dim3(blocks_for(np)), dim3(256), 0, st, (long long)n, S->start_key_sorted.as<uint32_t>(),
                     S->start_seq_sorted.as<uint32_t>(), S->start_val.as<unsigned long long>(),
                     (const unsigned long long*)I->dev.start_set, S->ray_flags.as<uint32_t>());
  VGX_HIP(ctx, hipGetLastError());
  // ---- 2. walks written out (bounded speculation: see det_count_kernel), sorted, swept to the fixed point ----
  // The first count is of the COMPLETE walks (det_points_kernel marked every ray "written out completely").  Small
  // scans (a LiDAR sweep at 0.2 m: < 1 M steps) are swept as they are -- an attempt costs a dozen launches and a
  // read-back, more than the steps saved; large ones (a depth image at 0.05 m: 18 M steps) are cut to `det_cap`
  // (32) steps per ray and extended on demand.  With the early-out switched off every ray runs its full length
  // anyway.  Once a scan had to be cut, the next ones are cut from the start (S->start_capped), with the marks of
  // the rays that ran on kept from scan to scan (det_extend_kernel).
  bool capped = start_capped;  // this attempt cut rays short: whether one of them ran on has to be looked at afterwards
  bool may_cap = may_cap_at_all && !start_capped;
  // (experiment aids: VGX_DET_CAP / VGX_DET_CAP_THRESHOLD override the integrator's speculation depth / threshold)
  static const long long cap_env = getenv("VGX_DET_CAP") ? atoll(getenv("VGX_DET_CAP")) : 0;
  static const long long thr_env = getenv("VGX_DET_CAP_THRESHOLD") ? atoll(getenv("VGX_DET_CAP_THRESHOLD")) : -1;
  const uint32_t kCapThreshold = thr_env >= 0 ? (uint32_t)std::min<long long>(thr_env, 0xffffffffll) : I->det_cap_threshold;
  const uint32_t cap = std::max<uint32_t>(cap_env > 0 ? (uint32_t)cap_env : I->det_cap, 1u);
  static const int mark_life = getenv("VGX_DET_MARK_LIFE") ? std::max(1, std::min(200, atoi(getenv("VGX_DET_MARK_LIFE")))) : 2;  // (depth image, ms per scan: 1 -> 1.25, 2 -> 1.17, 4 -> 1.18, 8 -> 1.21)
  static const bool scan_sweeps = getenv("VGX_DET_SWEEP") && !strcmp(getenv("VGX_DET_SWEEP"), "scan");  // A/B aid
  size_t N = 0;
  int walks_done = 0;
  bool range_error = false;
  HappenedOp happened{nullptr, nullptr, nullptr};
  UpdateOp update{nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, 0u};
  for (int attempt = 0;; ++attempt) {
    if (attempt > 64) return set_error(ctx, VGX_ERR_HIP, "TSDF reproducible mode: speculation did not settle (internal error)");
    {
      const uint32_t count_tiles = blocks_for(np + 1);
      TileChain chain;
      DET_TRY(next_chain(ctx, S, count_tiles, &chain));
      hipLaunchKernelGGL(det_count_kernel, dim3(count_tiles), dim3(256), 0, st, c, vsi, T[4], T[5], T[6], (long long)n,
                         S->ray_pg.as<float4>(), S->ray_flags.as<uint32_t>(), S->ext.as<uint8_t>(), S->count.as<uint32_t>(),
                         S->full_count.as<uint32_t>(), S->start_key_sorted.as<uint32_t>(), S->start_seq_sorted.as<uint32_t>(),
                         S->start_val.as<unsigned long long>(), I->dev.start_set, cap, S->d_ctr, chain, S->off.as<uint32_t>());
      VGX_HIP(ctx, hipGetLastError());
    }
    DET_TRY(read_counters(ctx, S));
    pc.mark(0);
    ++pc.attempts;
    if (S->h_ctr[kCtrError]) return det_error_code(ctx, S->h_ctr[kCtrError]);
    const unsigned long long total = S->h_ctr[kCtrTotal];
    if (may_cap) {
      may_cap = false;
      if (total > kCapThreshold) {  // count again, this time cut to the cap
        VGX_HIP(ctx, hipMemsetAsync(S->ext.p, 0, np + 1, st));
        VGX_HIP(ctx, hipMemsetAsync(S->d_ctr + kCtrTotal, 0, 8, st));
        capped = true;
        S->start_capped = true;
        continue;
      }
    } else if (attempt == 0 && start_capped && total < kCapThreshold / 16) {
      S->start_capped = false;  // little even when cut: the next scan looks at its complete walks first again
    }
    if (total >= (1ull << 32) - 2)  // (the offsets are 32 bits wide)
      return set_error(ctx, VGX_ERR_UNSUPPORTED, "TSDF reproducible mode: more than 2^32 voxel steps in a scan");
    N = (size_t)total;
    if (N == 0) return VGX_OK;  // nothing was cast (the start set has been updated)
    DET_TRY(grow(ctx, S->acc_vox, N * 8));
    DET_TRY(grow(ctx, S->acc_key, (N + 1) * 4));  // + 1: reused for the compaction offsets
    DET_TRY(grow(ctx, S->acc_ray, N * 4));
    DET_TRY(grow(ctx, S->s_key, N * 4));
    DET_TRY(grow(ctx, S->s_idx, N * 4));
    DET_TRY(grow(ctx, S->s_r, N * 4));
    DET_TRY(grow(ctx, S->s_k, N * 4));
    DET_TRY(grow(ctx, S->s_h, N * 4));
    DET_TRY(grow(ctx, S->last, (N + 1) * 4));
    DET_TRY(grow(ctx, S->seen, N + 16));  // (+ 16: det_ray_kernel reads whole words around a ray's flags)
    DET_TRY(grow(ctx, S->c_idx, N * 4));
    DET_TRY(grow(ctx, S->c_key, N * 4));
    hipLaunchKernelGGL(det_walk_kernel, dim3(blocks_for(np)), dim3(256), 0, st, c, vsi, T[4], T[5], T[6], (long long)n,
                       S->ray_pg.as<float4>(), S->ray_flags.as<uint32_t>(), S->count.as<uint32_t>(), S->off.as<uint32_t>(),
                       I->dev.observed_offset, S->acc_vox.as<unsigned long long>(), S->acc_key.as<uint32_t>(),
                       S->acc_ray.as<uint32_t>(), S->T.as<int32_t>(), S->broke.as<uint8_t>(), S->walked.as<uint32_t>(),
                       walks_done > 0 ? 1 : 0, S->d_ctr);
    VGX_HIP(ctx, hipGetLastError());
    ++walks_done;
    // stable: inside a slot the accesses stay in (ray, step) = visiting order
    DET_TRY(sort_by_slot(ctx, S, S->acc_key.as<uint32_t>(), S->s_key.as<uint32_t>(), S->s_idx.as<uint32_t>(), N, kSetBits));
    hipLaunchKernelGGL(det_gather_kernel, dim3(blocks_for(N)), dim3(256), 0, st, N, S->s_idx.as<uint32_t>(),
                       S->acc_vox.as<unsigned long long>(), S->acc_ray.as<uint32_t>(), S->off.as<uint32_t>(),
                       S->s_r.as<uint32_t>(), S->s_k.as<uint32_t>(), S->s_h.as<uint32_t>());
    VGX_HIP(ctx, hipGetLastError());
    // ---- sweeps to the fixed point ----
    happened = HappenedOp{S->s_r.as<uint32_t>(), S->s_k.as<uint32_t>(), S->T.as<int32_t>()};
    update = UpdateOp{S->s_r.as<uint32_t>(), S->s_k.as<uint32_t>(), S->T.as<int32_t>(), S->broke.as<uint8_t>(),
                      nullptr, nullptr, (uint32_t)N};
    auto pos = rocprim::make_counting_iterator<uint32_t>(0u);
    if (scan_sweeps) {  // (A/B aid: the sweep over a rocprim scan)
      size_t scan_bytes = 0;
      VGX_HIP(ctx, rocprim::exclusive_scan(nullptr, scan_bytes, rocprim::make_transform_iterator(pos, happened),
                                           S->last.as<uint32_t>(), 0u, N, rocprim::maximum<uint32_t>(), st));
      DET_TRY(grow(ctx, S->tmp, scan_bytes));
    }
    const uint32_t tiles = (uint32_t)((N + kSweepTile - 1) / kSweepTile);
    DET_TRY(grow_zeroed(ctx, S->tile_state, (size_t)tiles * 8));
    {
      // A sweep that changes no stopping step is the fixed point (and so is every sweep after it); n + 1 sweeps
      // always suffice.  A sweep is two launches -- det_sweep_kernel, det_ray_kernel -- and the ray kernel's last
      // block writes {sweep number, "something moved"} into a pinned word.  The host never waits for the stream:
      // it keeps `ahead` sweeps queued beyond the one it is waiting to hear from and polls the word (round 3: a
      // copy + stream synchronise every look, 20-40 us each, and four launches a sweep).  Sweeps past the fixed
      // point change nothing, so what was queued in vain costs their run time only: on small scans (a sweep over
      // < 1 M accesses is ~10 us, less than the bubble of waiting) two are kept queued, on large ones none.
      static const int ahead_env = getenv("VGX_DET_AHEAD") ? atoi(getenv("VGX_DET_AHEAD")) : -1;  // experiment aid
      const int ahead = ahead_env >= 0 ? ahead_env : N <= (1u << 20) ? 2 : 0;
      const unsigned long long seq0 = S->flag_seq;  // sweep j of this attempt reports as seq0 + j + 1
      long long issued = 0, heard = -1;              // sweeps launched; the latest sweep the host has heard from
      bool settled = false;
      struct Tally {  // (experiment aid)
        PhaseClock& pc;
        long long& issued;
        ~Tally() { pc.sweeps += issued; }
      } tally{pc, issued};
      while (!settled) {
        if (issued > n + 2 + ahead + (walks_done > 1 ? (long long)N : 0))  // (a warm start: at least a step a sweep)
          return set_error(ctx, VGX_ERR_HIP, "TSDF reproducible mode: the sweeps did not settle (internal error)");
        while (issued <= heard + 1 + ahead) {
          if (scan_sweeps) {
            size_t bytes = S->tmp.bytes;
            VGX_HIP(ctx, rocprim::exclusive_scan(S->tmp.p, bytes, rocprim::make_transform_iterator(pos, happened),
                                                 S->last.as<uint32_t>(), 0u, N, rocprim::maximum<uint32_t>(), st));
            hipLaunchKernelGGL(det_seen_kernel, dim3(blocks_for(N)), dim3(256), 0, st, N, S->s_key.as<uint32_t>(),
                               S->s_idx.as<uint32_t>(), S->s_h.as<uint32_t>(), S->last.as<uint32_t>(), I->dev.observed_set,
                               I->dev.observed_offset, S->seen.as<uint8_t>());
          } else {
            if (S->sweep_epoch >= kSweepEpochMax) {  // (after 2^30 sweeps: start the tags over)
              VGX_HIP(ctx, hipMemsetAsync(S->tile_state.p, 0, S->tile_state.bytes, st));
              S->sweep_epoch = 0;
            }
            ++S->sweep_epoch;
            hipLaunchKernelGGL(det_sweep_kernel, dim3(tiles), dim3(256), 0, st, (uint32_t)N, (unsigned long long)S->sweep_epoch,
                               S->d_ctr, S->tile_state.as<unsigned long long>(), S->s_key.as<uint32_t>(),
                               S->s_idx.as<uint32_t>(), S->s_h.as<uint32_t>(), S->s_r.as<uint32_t>(), S->s_k.as<uint32_t>(),
                               S->T.as<int32_t>(), I->dev.observed_set, I->dev.observed_offset, S->seen.as<uint8_t>(),
                               S->last.as<uint32_t>());
          }
          VGX_HIP(ctx, hipGetLastError());
          ++S->flag_seq;
          hipLaunchKernelGGL(det_ray_kernel, dim3(blocks_for(np)), dim3(256), 0, st, (long long)n,
                             (int)c.max_consecutive_ray_collisions, S->count.as<uint32_t>(), S->off.as<uint32_t>(),
                             S->seen.as<uint8_t>(), S->T.as<int32_t>(), S->broke.as<uint8_t>(), S->d_ctr, S->flag_seq,
                             S->d_flag);
          VGX_HIP(ctx, hipGetLastError());
          ++issued;
        }
        unsigned long long word = 0;
        DET_TRY(wait_for_sweep(ctx, S, seq0 + (unsigned long long)heard + 2ull, &word));  // ... at least the next one
        pc.mark(heard < 0 ? 1 : 2);
        heard = (long long)((word >> 2) - seq0) - 1;
        if (word & 2ull) range_error = true;
        if (range_error || !(word & 1ull)) settled = true;
      }
    }
    if (range_error) break;
    if (!capped) break;  // every ray was written out completely: nothing can have been cut short
    hipLaunchKernelGGL(det_extend_kernel, dim3(blocks_for(np)), dim3(256), 0, st, (long long)n, S->count.as<uint32_t>(),
                       S->full_count.as<uint32_t>(), S->broke.as<uint8_t>(), S->T.as<int32_t>(), S->ext.as<uint8_t>(), cap, mark_life, S->d_ctr);
    VGX_HIP(ctx, hipGetLastError());
    DET_TRY(read_counters(ctx, S));
    pc.mark(3);
    if (S->h_ctr[kCtrError]) {
      range_error = true;
      break;
    }
    if (!S->h_ctr[kCtrOverflow]) {  // no ray was cut short: this is the sequential execution
      S->ext_points = n;
      break;
    }
  }
  if (range_error) {
    DET_TRY(read_counters(ctx, S));  // (also: what the sweeps queued ahead were still writing is off the stream)
    return det_error_code(ctx, S->h_ctr[kCtrError] ? S->h_ctr[kCtrError] : 2ull);
  }
  return det_commit(I, S, T, N, &update, true, true, S->ray_pg.as<float4>(), S->ray_color.as<uint32_t>(), n_updates,
                    &pc);
}

int det_merged_commit(vgx_tsdf_integrator I, const float T[7], long long n, const float4* g_pg, const uint32_t* g_color,
                      const uint32_t* g_flags, uint32_t* g_count, const unsigned long long* keys_sorted,
                      const unsigned int* group_start, const unsigned int* counters, int64_t* n_updates) {
  vgx_ctx ctx = I->ctx;
  const vgx_tsdf_config& c = I->dev.cfg;
  hipStream_t st = ctx->tsdf_stream;
  if (n_updates) *n_updates = 0;
  DET_TRY(ensure_scratch(I));
  DetScratch* S = I->det;
  // g_count is zero beyond the last group (the caller cleared it): the scan runs over n + 1 entries and
  // the number of groups never has to come to the host
  const size_t G = (size_t)n;
  // (the counters were zeroed by merged_bundle_kernel; merged_merge_kernel left the number of ray steps -- 64 bits:
  // the 32-bit offsets below wrap silently -- and its error code in them: ONE copy is the scan's whole read-back)
  DET_TRY(grow(ctx, S->off, (G + 1) * 4));
  {
    const uint32_t scan_tiles = (uint32_t)((G + 1 + 256 * kScanIpt - 1) / (256 * kScanIpt));
    TileChain chain;
    DET_TRY(next_chain(ctx, S, scan_tiles, &chain));
    hipLaunchKernelGGL(det_offsets_kernel, dim3(scan_tiles), dim3(256), 0, st, (const uint32_t*)g_count, S->off.as<uint32_t>(),
                       (uint32_t)(G + 1), chain);
    VGX_HIP(ctx, hipGetLastError());
  }
  DET_TRY(read_counters(ctx, S));
  // (merged_bundle_kernel: the key corner; merged_merge_kernel: a ray too long; the chained prefix sums: 3)
  if (S->h_ctr[kCtrError]) return det_error_code(ctx, S->h_ctr[kCtrError]);
  if (S->h_ctr[kCtrTotal] >= (1ull << 32) - 2)
    return set_error(ctx, VGX_ERR_UNSUPPORTED, "TSDF merged integrator: more than 2^32 voxel steps in a scan");
  const uint32_t total = (uint32_t)S->h_ctr[kCtrTotal];
  {
    // lanes per group for the next scan's merge kernel: about as many as a group has points
    const unsigned long long groups = S->h_ctr[kCtrGroups], points = S->h_ctr[kCtrGroupPoints];
    static const int forced = getenv("VGX_MERGED_LANES") ? atoi(getenv("VGX_MERGED_LANES")) : 0;  // experiment aid
    if (forced == 4 || forced == 8 || forced == 16) I->merged_lanes = forced;
    // (measured, ms per scan with 4 / 8 / 16 lanes: depth image, ~8 points a group, 0.794 / 0.812 / 0.852; LiDAR, one or
    // two, 0.210 / 0.208 / 0.207 -- wide groups only pay where a group's points outnumber them several times)
    else if (groups > 0) I->merged_lanes = points <= 16 * groups ? 4 : points <= 48 * groups ? 8 : 16;
  }
  const size_t N = total;
  if (N == 0) return VGX_OK;
  DET_TRY(grow(ctx, S->acc_vox, N * 8));
  DET_TRY(grow(ctx, S->acc_key, (N + 1) * 4));
  DET_TRY(grow(ctx, S->acc_ray, N * 4));
  DET_TRY(grow(ctx, S->s_key, N * 4));
  DET_TRY(grow(ctx, S->s_idx, N * 4));
  DET_TRY(grow(ctx, S->s_h, N * 4));
  DET_TRY(grow(ctx, S->last, (N + 1) * 4));
  DET_TRY(grow(ctx, S->seen, N));
  DET_TRY(grow(ctx, S->c_idx, N * 4));
  DET_TRY(grow(ctx, S->c_key, N * 4));
  const unsigned walk_grid = (unsigned)std::min<size_t>((G * 16 + 255) / 256, (size_t)ctx->cu_count * 16);
  hipLaunchKernelGGL(det_merged_walk_kernel, dim3(walk_grid), dim3(256), 0, st, c, I->layer->dev.voxel_size_inv, T[4], T[5],
                     T[6], (uint32_t)G, g_pg, g_flags, g_count, S->off.as<uint32_t>(), keys_sorted, group_start, counters,
                     (int)c.enable_anti_grazing, S->acc_vox.as<unsigned long long>(),
                     S->acc_key.as<uint32_t>(), S->acc_ray.as<uint32_t>(), S->seen.as<uint8_t>(), S->d_ctr);
  VGX_HIP(ctx, hipGetLastError());
  // stable: a voxel's updates stay in (group, step) order
  DET_TRY(sort_by_slot(ctx, S, S->acc_key.as<uint32_t>(), S->s_key.as<uint32_t>(), S->s_idx.as<uint32_t>(), N, kSetBits));
  {
    // Every voxel a ray visits lies between the sensor and its (clipped) end point plus the truncation
    // band: when that box is inside +-2^20 voxels the walk kernel's range flag cannot be set and does not
    // have to be waited for (one stream round trip per scan).
    const double reach = (double)c.max_ray_length_m + 2.0 * (double)c.default_truncation_distance + 2.0 * I->layer->dev.voxel_size;
    const double far = std::max(std::max(std::fabs((double)T[4]), std::fabs((double)T[5])), std::fabs((double)T[6])) + reach;
    if (!(far * (double)I->layer->dev.voxel_size_inv + 4.0 < (double)kVoxBias)) {
      DET_TRY(read_counters(ctx, S));
      if (S->h_ctr[kCtrError]) return det_error_code(ctx, S->h_ctr[kCtrError]);
    }
  }
  // only anti-grazing removes updates; blocks in order of first update only in the reproducible mode
  // (otherwise on demand: the VALUES are order-exact either way, only the pool order is arrival order)
  const UpdateOp update{nullptr, nullptr, nullptr, nullptr, S->s_idx.as<uint32_t>(), S->seen.as<uint8_t>(), (uint32_t)N};
  return det_commit(I, S, T, N, c.enable_anti_grazing ? &update : nullptr, false, c.deterministic != 0, g_pg,
                    g_color, n_updates);
}

}  // namespace vgx
