// Stable sort of (key, position) pairs for the sort-based TSDF paths (vgx_tsdf_det.hip: points by start-set slot,
// speculative accesses by observed-set slot -- the orders voxblox's single-threaded integrators produce by
// visiting one ray after the other; call site voxgraph/src/frontend/measurement_processors/
// pointcloud_integrator.cpp:83).  gfx950 only.
//
// Why not rocprim's: a scan is 10^4 .. 10^6.5 records with 20-21 key bits.  Below 2^20 items rocprim sorts by
// merging (7-17 launches of 5-10 us each for a LiDAR scan's records, vgx_tsdf_internal.h FewPassSort), above by
// onesweep with two fills a pass (11 launches).  At these sizes a launch is its latency -- the chain of dependent
// memory round trips inside it -- so what counts is the number of launches and of hops in each.  Here: ONE
// counting launch + one launch per 8 key bits (LSD: 3 passes for 17-24 bits), tiles of 4096 records, nothing to
// clear in between.
//   Ranks inside a tile (both schemes): a wave takes 1024 consecutive records in 16 rounds of 64; in a round,
// lanes with the same digit find each other with eight ballots, the first of them bumps the wave's counter of
// that digit (LDS), everybody adds the number of same-digit lanes below it.  Record order = (wave, round, lane)
// = position, so equal digits keep their order: stable.
//   Where a tile's records of digit d go = records of smaller digits + records of digit d in earlier tiles:
//   * up to 128 tiles (524 288 records: every LiDAR scan, a depth image's points): a count MATRIX [tile][digit]
//     per pass.  The counting launch stores pass 0's rows (a tile's own histogram) and zeroes the others; a pass
//     reads its whole column block -- `tiles` independent loads per thread, ONE round trip, no waiting for other
//     workgroups at all -- and, as it scatters, counts every record into the row of the tile it lands in for the
//     next pass (global atomics).  Two dependent hops per launch.
//   * above: a CHAIN.  Tiles are taken in the order the workgroups start (a ticket: a tile only ever waits for
//     tiles already running), the counts are chained by 8-byte words {epoch, status, value} per (tile, digit),
//     tagged with the launch's epoch (words of earlier launches read as "not there yet", nothing is cleared);
//     thread d owns digit d: publishes the tile's count, walks back over the tiles before it, four words in
//     flight, until it meets an inclusive prefix, publishes its own.  Digit totals come from a global histogram
//     the counting launch builds; the last workgroup of the last pass zeroes it and the tickets again.
#include <algorithm>
#include <cstdint>
#include <cstdlib>
#include <cstring>

#include "vgx_tsdf_internal.h"
#include "voxgraph_amd_bench.h"  // vgx_bench_slot_sort

namespace vgx {

namespace {

constexpr int kSortIpt = 16;                       // records per thread
constexpr uint32_t kSortTile = 256u * kSortIpt;    // per workgroup
constexpr int kSortMaxPasses = 4;
constexpr uint32_t kSortMatrixTiles = 128;         // the count-matrix scheme up to here (tiles^2 KB of column reads)
// chain scheme, the counter block (u32): tickets per pass, arrivals of the last pass, histograms [pass][digit]
enum { kSortTicket = 0, kSortArrive = kSortMaxPasses, kSortHist = 8, kSortWords = kSortHist + kSortMaxPasses * 256 };
constexpr unsigned long long kSortEpochMax = (1ull << 30) - 1;

// chain scheme: digit totals of every pass
__global__ __launch_bounds__(256) void slot_sort_hist_kernel(const uint32_t* __restrict__ keys, uint32_t n, int passes,
                                                            uint32_t* __restrict__ ctr) {
  __shared__ uint32_t h[kSortMaxPasses][256];
  for (int p = 0; p < kSortMaxPasses; ++p) h[p][threadIdx.x] = 0u;
  __syncthreads();
  for (uint32_t i = blockIdx.x * 256u + threadIdx.x; i < n; i += gridDim.x * 256u) {
    const uint32_t key = keys[i];
    for (int p = 0; p < passes; ++p) atomicAdd(&h[p][(key >> (8 * p)) & 255u], 1u);
  }
  __syncthreads();
  for (int p = 0; p < passes; ++p) {
    const uint32_t v = h[p][threadIdx.x];
    if (v) atomicAdd(&ctr[kSortHist + p * 256 + threadIdx.x], v);
  }
}

// matrix scheme: workgroup t stores row t of pass 0's matrix (the histogram of its own tile's lowest digit) and
// zeroes row t of the later passes' matrices (they are filled by atomics while the pass before them scatters)
__global__ __launch_bounds__(256) void slot_sort_rows_kernel(const uint32_t* __restrict__ keys, uint32_t n, int passes,
                                                            uint32_t tiles_cap, uint32_t* __restrict__ matrix) {
  __shared__ uint32_t h[256];
  h[threadIdx.x] = 0u;
  __syncthreads();
  const uint32_t base = blockIdx.x * kSortTile + threadIdx.x;
#pragma unroll
  for (int j = 0; j < kSortIpt; ++j) {
    const uint32_t i = base + 256u * j;
    if (i < n) atomicAdd(&h[keys[i] & 255u], 1u);
  }
  __syncthreads();
  uint32_t* row = matrix + (size_t)blockIdx.x * 256u + threadIdx.x;
  row[0] = h[threadIdx.x];
  for (int p = 1; p < passes; ++p) row[(size_t)p * tiles_cap * 256u] = 0u;
}

__device__ __forceinline__ unsigned long long state_load(const unsigned long long* p) {
  return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ void state_store(unsigned long long* p, unsigned long long v) {
  __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// keys_in / vals_in (nullptr: a record's value is its position) -> keys_out / vals_out, stable by bits
// [8 pass, 8 pass + 8) of the key.  MATRIX: `table` = this pass's count matrix [tiles][256], `next_table` the
// next pass's (nullptr on the last pass); else `table` = the counter block, `state` the chain words.
template <bool MATRIX>
__global__ __launch_bounds__(256) void slot_sort_pass_kernel(const uint32_t* __restrict__ keys_in,
                                                            const uint32_t* __restrict__ vals_in,
                                                            uint32_t* __restrict__ keys_out, uint32_t* __restrict__ vals_out,
                                                            uint32_t n, int pass, int last_pass, unsigned long long epoch,
                                                            uint32_t* __restrict__ table, uint32_t* __restrict__ next_table,
                                                            unsigned long long* __restrict__ state,
                                                            unsigned long long* __restrict__ error_word) {
  __shared__ uint32_t wave_digit[4][256];  // per wave and digit: records counted so far -> where its records go
  __shared__ uint32_t sh_tile, sh_last, sh_scan[4];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int shift = 8 * pass;
  if (!MATRIX && tid == 0) sh_tile = atomicAdd(&table[kSortTicket + pass], 1u);
#pragma unroll
  for (int w = 0; w < 4; ++w) wave_digit[w][tid] = 0u;
  __syncthreads();
  const uint32_t tile = MATRIX ? blockIdx.x : sh_tile;
  const uint32_t first = tile * kSortTile + (uint32_t)wave * (64u * kSortIpt) + (uint32_t)lane;
  uint32_t key[kSortIpt], rank[kSortIpt];
#pragma unroll
  for (int j = 0; j < kSortIpt; ++j) {
    const uint32_t i = first + 64u * j;
    key[j] = i < n ? keys_in[i] : 0xffffffffu;
  }
  // MATRIX: digit `tid` in the tiles before this one, and in all of them (independent loads, next to the keys')
  uint32_t prefix = 0, total_d = 0;
  if (MATRIX) {
    const uint32_t tiles = gridDim.x;
    for (uint32_t t0 = 0; t0 < tiles; t0 += 8) {
      uint32_t c[8];
#pragma unroll
      for (int q = 0; q < 8; ++q) c[q] = t0 + q < tiles ? table[(size_t)(t0 + q) * 256u + tid] : 0u;
#pragma unroll
      for (int q = 0; q < 8; ++q) {
        total_d += c[q];
        prefix += t0 + q < tile ? c[q] : 0u;
      }
    }
  }
  const unsigned long long below_me = __lanemask_lt();
#pragma unroll
  for (int j = 0; j < kSortIpt; ++j) {
    const bool valid = first + 64u * j < n;
    const uint32_t d = (key[j] >> shift) & 255u;
    unsigned long long same = __ballot(valid);  // the lanes of this round with my digit
#pragma unroll
    for (int b = 0; b < 8; ++b) {
      const bool bit = (d >> b) & 1u;
      const unsigned long long with = __ballot(bit);
      same &= bit ? with : ~with;
    }
    const int leader = same ? __ffsll((long long)same) - 1 : 0;
    uint32_t before = 0;
    if (valid && lane == leader) before = atomicAdd(&wave_digit[wave][d], (uint32_t)__popcll(same));
    before = (uint32_t)__shfl((int)before, leader);
    rank[j] = before + (uint32_t)__popcll(same & below_me);
  }
  __syncthreads();
  // thread d owns digit d from here
  uint32_t wave_off[4], mine = 0;
#pragma unroll
  for (int w = 0; w < 4; ++w) {
    wave_off[w] = mine;
    mine += wave_digit[w][tid];
  }
  // where digit d starts in the output: the exclusive scan of the digit totals
  if (!MATRIX) total_d = table[kSortHist + pass * 256 + tid];
  uint32_t inc = total_d;
#pragma unroll
  for (int s = 1; s < 64; s <<= 1) {
    const uint32_t o = (uint32_t)__shfl_up((int)inc, s);
    if (lane >= s) inc += o;
  }
  if (lane == 63) sh_scan[wave] = inc;
  __syncthreads();
  uint32_t digit_base = inc - total_d;
#pragma unroll
  for (int w = 0; w < 4; ++w)
    if (w < wave) digit_base += sh_scan[w];
  if (!MATRIX) {
    // the records of digit d in the tiles before this one
    const unsigned long long tag = epoch << 34;
    unsigned long long* my_word = state + (size_t)tile * 256u + tid;
    state_store(my_word, tag | ((tile == 0 ? 2ull : 1ull) << 32) | mine);
    bool failed = false;
    for (uint32_t t = tile; t > 0 && !failed;) {
      const int look = (int)min(t, 4u);
      unsigned long long x[4];
#pragma unroll
      for (int q = 0; q < 4; ++q)
        if (q < look) x[q] = state_load(state + (size_t)(t - 1 - q) * 256u + tid);
      bool done = false;
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        if (q >= look || done || failed) continue;
        unsigned spins = 0;
        while ((x[q] >> 34) != epoch) {
          __builtin_amdgcn_s_sleep(1);
          if (++spins > (1u << 22)) {  // (seconds: a tile that started before this one never reported -- internal error)
            failed = true;
            break;
          }
          x[q] = state_load(state + (size_t)(t - 1 - q) * 256u + tid);
        }
        if (failed) break;
        prefix += (uint32_t)x[q];
        if (((x[q] >> 32) & 3ull) == 2ull) done = true;
      }
      if (done) break;
      t -= (uint32_t)look;
    }
    if (failed && error_word) *error_word = 3ull;
    if (tile != 0) state_store(my_word, tag | (2ull << 32) | (unsigned long long)(prefix + mine));
  }
#pragma unroll
  for (int w = 0; w < 4; ++w) wave_digit[w][tid] = digit_base + prefix + wave_off[w];
  __syncthreads();
#pragma unroll
  for (int j = 0; j < kSortIpt; ++j) {
    const uint32_t i = first + 64u * j;
    if (i < n) {
      const uint32_t pos = wave_digit[wave][(key[j] >> shift) & 255u] + rank[j];
      if (pos < n) {  // (always, unless a look back failed)
        keys_out[pos] = key[j];
        vals_out[pos] = vals_in ? vals_in[i] : i;
        if (MATRIX && next_table) atomicAdd(&next_table[(size_t)(pos / kSortTile) * 256u + ((key[j] >> (shift + 8)) & 255u)], 1u);
      }
    }
  }
  if (MATRIX || !last_pass) return;
  // the sort's last workgroup leaves tickets and histograms at zero for the next sort
  __syncthreads();
  if (tid == 0) sh_last = atomicAdd(&table[kSortArrive], 1u) == gridDim.x - 1u ? 1u : 0u;
  __syncthreads();
  if (sh_last)
    for (int i = tid; i < kSortWords; i += 256) table[i] = 0u;
}

}  // namespace

struct SlotSort {
  uint32_t* d_matrix = nullptr;  // matrix scheme: [pass][kSortMatrixTiles][256]
  uint32_t* d_ctr = nullptr;     // chain scheme: the counter block ...
  unsigned long long* d_state = nullptr;  // ... and the chain words [tile][256]
  size_t state_tiles = 0;
  uint32_t* d_tmp = nullptr;  // keys then values of the intermediate pass
  size_t tmp_items = 0;
  uint32_t epoch = 0;
  bool clean = false;  // the counter block is all zero (false: a sort was cut short, or nothing is allocated yet)
};

void slot_sort_free(SlotSort* s) {
  if (!s) return;
  if (s->d_matrix) (void)hipFree(s->d_matrix);
  if (s->d_ctr) (void)hipFree(s->d_ctr);
  if (s->d_state) (void)hipFree(s->d_state);
  if (s->d_tmp) (void)hipFree(s->d_tmp);
  delete s;
}

bool slot_sort_wanted(size_t n) {
  static const bool off = getenv("VGX_TSDF_SORT") && (!strcmp(getenv("VGX_TSDF_SORT"), "rocprim") || !strcmp(getenv("VGX_TSDF_SORT"), "default"));
  return !off && n > kSortTile && n < (1ull << 32) - kSortTile;  // (one tile or less: rocprim's single-block sort, one launch)
}

int slot_sort_pairs(vgx_ctx ctx, SlotSort** handle, const uint32_t* keys, uint32_t* keys_sorted, uint32_t* idx_sorted, size_t n,
                    unsigned end_bit, unsigned long long* error_word) {
  hipStream_t st = ctx->stream;
  if (!*handle) *handle = new (std::nothrow) SlotSort();
  SlotSort* S = *handle;
  if (!S) return set_error(ctx, VGX_ERR_NOMEM, "TSDF sort: out of host memory");
  const int passes = (int)std::min<unsigned>((end_bit + 7u) / 8u, (unsigned)kSortMaxPasses);
  const uint32_t tiles = (uint32_t)((n + kSortTile - 1) / kSortTile);
  static const bool chain_only = getenv("VGX_TSDF_SORT") && !strcmp(getenv("VGX_TSDF_SORT"), "chain");  // A/B aid
  const bool matrix = tiles <= kSortMatrixTiles && !chain_only;
  if (passes > 1 && S->tmp_items < n) {
    VGX_HIP(ctx, hipStreamSynchronize(st));
    if (S->d_tmp) (void)hipFree(S->d_tmp);
    S->d_tmp = nullptr;
    S->tmp_items = 0;
    const size_t want = n + n / 4 + 1024;
    VGX_HIP(ctx, hipMalloc(&S->d_tmp, want * 8));
    S->tmp_items = want;
  }
  if (matrix) {
    if (!S->d_matrix) VGX_HIP(ctx, hipMalloc(&S->d_matrix, (size_t)kSortMaxPasses * kSortMatrixTiles * 256 * 4));
    hipLaunchKernelGGL(slot_sort_rows_kernel, dim3(tiles), dim3(256), 0, st, keys, (uint32_t)n, passes, kSortMatrixTiles,
                       S->d_matrix);
    VGX_HIP(ctx, hipGetLastError());
  } else {
    if (!S->d_ctr) {
      VGX_HIP(ctx, hipMalloc(&S->d_ctr, kSortWords * 4));
      S->clean = false;
    }
    if (S->state_tiles < tiles) {
      VGX_HIP(ctx, hipStreamSynchronize(st));
      if (S->d_state) (void)hipFree(S->d_state);
      S->d_state = nullptr;
      S->state_tiles = 0;
      const size_t want = (size_t)tiles + tiles / 4 + 16;
      VGX_HIP(ctx, hipMalloc(&S->d_state, want * 256 * 8));
      VGX_HIP(ctx, hipMemsetAsync(S->d_state, 0, want * 256 * 8, st));  // (no tag of a launch to come)
      S->state_tiles = want;
      S->epoch = 0;
    }
    if (!S->clean) VGX_HIP(ctx, hipMemsetAsync(S->d_ctr, 0, kSortWords * 4, st));
    S->clean = false;
    const unsigned hist_grid = (unsigned)std::min<uint32_t>(tiles, (uint32_t)ctx->cu_count * 4u);
    hipLaunchKernelGGL(slot_sort_hist_kernel, dim3(hist_grid), dim3(256), 0, st, keys, (uint32_t)n, passes, S->d_ctr);
    VGX_HIP(ctx, hipGetLastError());
  }
  const uint32_t* src_k = keys;
  const uint32_t* src_v = nullptr;
  for (int p = 0; p < passes; ++p) {
    const bool to_out = ((passes - 1 - p) & 1) == 0;  // the last pass lands in the output
    uint32_t* dst_k = to_out ? keys_sorted : S->d_tmp;
    uint32_t* dst_v = to_out ? idx_sorted : S->d_tmp + S->tmp_items;
    const int last = p == passes - 1 ? 1 : 0;
    if (matrix) {
      uint32_t* table = S->d_matrix + (size_t)p * kSortMatrixTiles * 256;
      hipLaunchKernelGGL(slot_sort_pass_kernel<true>, dim3(tiles), dim3(256), 0, st, src_k, src_v, dst_k, dst_v, (uint32_t)n, p, last,
                         0ull, table, last ? (uint32_t*)nullptr : table + (size_t)kSortMatrixTiles * 256,
                         (unsigned long long*)nullptr, error_word);
    } else {
      if (S->epoch >= kSortEpochMax) {  // (after 2^30 passes: start the tags over)
        VGX_HIP(ctx, hipMemsetAsync(S->d_state, 0, S->state_tiles * 256 * 8, st));
        S->epoch = 0;
      }
      ++S->epoch;
      hipLaunchKernelGGL(slot_sort_pass_kernel<false>, dim3(tiles), dim3(256), 0, st, src_k, src_v, dst_k, dst_v, (uint32_t)n, p, last,
                         (unsigned long long)S->epoch, S->d_ctr, (uint32_t*)nullptr, S->d_state, error_word);
    }
    VGX_HIP(ctx, hipGetLastError());
    src_k = dst_k;
    src_v = dst_v;
  }
  if (!matrix) S->clean = true;
  return VGX_OK;
}

}  // namespace vgx

using namespace vgx;

// test / bench tooling (include/voxgraph_amd_bench.h): the sort by itself, `repeats` times on one scratch object
extern "C" int vgx_bench_slot_sort(vgx_ctx ctx, const void* d_keys, int64_t n, int32_t end_bit, int32_t repeats,
                                   void* d_keys_sorted, void* d_idx_sorted, float* ms_per_sort) {
  if (!ctx || !d_keys || !d_keys_sorted || !d_idx_sorted || n <= 0 || n >= (1ll << 32) || end_bit <= 0 || end_bit > 32 || repeats <= 0)
    return VGX_ERR_INVALID;
  std::lock_guard<std::mutex> lk(ctx->mu);
  VGX_HIP(ctx, hipSetDevice(ctx->device));
  SlotSort* S = nullptr;
  unsigned long long* d_err = nullptr;
  VGX_HIP(ctx, hipMalloc(&d_err, 8));
  hipEvent_t e[2];
  for (auto& ev : e) (void)hipEventCreate(&ev);
  int rc = hipMemsetAsync(d_err, 0, 8, ctx->stream) == hipSuccess ? VGX_OK : VGX_ERR_HIP;
  // (the first sort allocates; it is timed apart from the rest)
  if (rc == VGX_OK)
    rc = slot_sort_pairs(ctx, &S, (const uint32_t*)d_keys, (uint32_t*)d_keys_sorted, (uint32_t*)d_idx_sorted, (size_t)n,
                         (unsigned)end_bit, d_err);
  (void)hipEventRecord(e[0], ctx->stream);
  for (int r = 1; r < repeats && rc == VGX_OK; ++r)
    rc = slot_sort_pairs(ctx, &S, (const uint32_t*)d_keys, (uint32_t*)d_keys_sorted, (uint32_t*)d_idx_sorted, (size_t)n,
                         (unsigned)end_bit, d_err);
  (void)hipEventRecord(e[1], ctx->stream);
  unsigned long long err = 0;
  float ms = 0.0f;
  if (rc == VGX_OK && (hipEventSynchronize(e[1]) != hipSuccess || hipEventElapsedTime(&ms, e[0], e[1]) != hipSuccess ||
                       hipMemcpy(&err, d_err, 8, hipMemcpyDeviceToHost) != hipSuccess))
    rc = VGX_ERR_HIP;
  for (auto& ev : e) (void)hipEventDestroy(ev);
  (void)hipFree(d_err);
  slot_sort_free(S);
  if (rc == VGX_ERR_HIP) return set_error(ctx, rc, "vgx_bench_slot_sort: HIP failure");
  if (rc != VGX_OK) return rc;
  if (err) return set_error(ctx, VGX_ERR_HIP, "vgx_bench_slot_sort: a tile never reported (internal error)");
  if (ms_per_sort) *ms_per_sort = repeats > 1 ? ms / (float)(repeats - 1) : 0.0f;
  return VGX_OK;
}
