// TSDF path, racing mode (the default): voxblox::FastTsdfIntegrator::integratePointCloud as its worker threads run
// it -- every ray its own thread, rays racing on the two approximate sets and on the voxels -- laid out for a
// wavefront machine (round 5; the one-thread-per-point kernel it replaces is tsdf_integrate_kernel in vgx_tsdf.hip,
// still reachable with VGX_TSDF_KERNEL=v1 for A/B runs).
//
// What the one-thread-per-point kernel paid for (profiles/r04_sq_breakdown.json: waves wait on memory 81-91 % of
// their cycles; 52 us for a LiDAR scan whose longest chain of exchanges is 4 us):
//   * 95 % of the lanes die at the start-set test and the survivors stay scattered over 1024 wavefronts;
//   * a ray learns whether to take step k + 1 only when the exchange of step k is back: one round trip per voxel;
//   * neighbouring beams walk the SAME voxels in lock step, so their per-voxel compare-and-swaps fail against each
//     other and are retried one round trip at a time (2-4 rays per voxel and step).
// Here one workgroup takes 256 points through three phases:
//   1. START SET.  Every lane tests its start cell; lanes of a wavefront that hold the same cell as their left
//      neighbour do not exchange at all -- had they, right behind the run's first lane, they would have found their
//      own value: "already present", the result one serial order of voxblox's threads gives.  Survivors set up their
//      ray and are compacted (ballot + prefix) into a queue in LDS.
//   2. WALK.  Up to eight lanes per ray (four / two / one where a workgroup has more than 32 / 64 / 128 rays), lane j on
//      voxel step pos + j of the ray's DDA (each lane advances the ray's state j times: the same f32 additions in the same
//      order as a sequential walk).  The early-out needs more than max_consecutive_ray_collisions observed voxels IN A
//      ROW, so with a current run of c the next mc + 1 - c exchanges happen whatever they return: those lanes exchange
//      together -- one round trip for up to mc + 1 steps, and nothing is written that the sequential ray would not have
//      written.  The lanes behind them PEEK at their slot with a plain load; a peeked prefix that cannot contain the
//      stop is exchanged together with the unconditional steps of the next round (up to eight steps per round trip).
//      Only exchanges decide: a peek merely selects which exchanges to issue, so a slot that changes between peek and
//      exchange costs, at worst, a few exchanges behind the stop (a window of one round trip; counted, `overrun` in the
//      statistics).  A wavefront runs its rounds without waiting for the other three.
//   3. UPDATES.  Every voxel step that survives becomes a record {sdf, weight, colour} chained to its voxel in a hash
//      table in LDS.  When the workgroup's rays are done (or the table is nearly full) ONE lane per distinct voxel
//      looks the block up (allocating it if new), loads {distance, weight} and the colour, folds the chain over them
//      in registers -- updateTsdfVoxel for each record in turn -- and publishes with one compare-and-swap: rays of
//      one workgroup no longer collide on a voxel, only workgroups do (adjacent rings / image rows), and those are
//      not in lock step.
// Every ordering this produces is one voxblox's threads can produce, with one stated exception: the exchanges a ray
// issues in one round reach the L2 in no particular order, where a CPU thread's are sequentially consistent (another
// ray can see step k + 1 observed and step k not yet, for the ~100 ns between two arrivals).
// HBM is not what bounds this (a scan is a few MB).  Measured (profiles/r05_tsdf_racing.txt; 64 x 1024 LiDAR scan, kernel
// alone): launch + an almost empty kernel 7 us; + phase 1 10 us; + the walk 26 us; + the folds 29 us -- against 47-52 us
// for the one-thread-per-point kernel; a 640 x 480 depth image 50 us against 174.  What is left of the walk is the
// instruction stream of a round (~600 instructions and a dozen LDS trips around one memory round trip, one wavefront per
// SIMD: nothing hides any of it), not memory.
//
// This header holds the kernel itself as a template <STATS, TRACE>; vgx_tsdf_coop.hip instantiates the two shipped
// forms (TRACE = false) for libvoxgraph_amd.so, csrc/bench/vgx_tsdf_diag.hip the two event-logging forms for
// libvoxgraph_amd_bench.so (tests/test_tsdf_replay_gpu.py replays their logs through the oracle): ONE source, so what
// the replay proves about the logged kernel is proved about the shipped one up to timing.
#ifndef VGX_TSDF_COOP_KERNEL_H_
#define VGX_TSDF_COOP_KERNEL_H_
#include "vgx_tsdf_internal.h"

#pragma clang fp contract(off)

namespace vgx {

namespace {

constexpr int kLanesPerRay = 8;
constexpr int kMaxRecs = 768;                 // update records pending in LDS
constexpr int kTable = 1024;                  // voxel hash table (power of two, load factor <= 0.75)
static_assert(kTable == 0x400, "phase 3 keeps a flag above the slot bits of occ[]: occ[o] & 0x3ff");
constexpr int kFlushAt = kMaxRecs - 256;      // a round adds at most 256 records
constexpr unsigned long long kEmptyKey = ~0ull;
constexpr uint32_t kNil = 0xffffffffu;
constexpr int kBias = 1 << 20;                // 21 bits per axis in a voxel key

struct RayRec {           // a cast ray: as phase 1 leaves it, and as a pass of the walk hands it on to the next (52 bytes)
  int curr[3];
  uint32_t sign_carry;    // (sign + 1) of the three axes, two bits each (bits 0-5); observed voxels in a row so far << 8
  float t_next[3], t_step[3];
  uint32_t left_lo, left_hi;  // voxels of the walk not yet exchanged
  uint32_t point;         // the point the ray belongs to: its end point, weight and colour are made again from it when a group
                          // takes the ray (round 6: 24 bytes less per ray -- with them 256 rays + the tables take 40.5 KB of
                          // LDS and FOUR workgroups fit a CU's 160 KB instead of three, which a depth image's 1200 workgroups
                          // are short of)
};

struct UpdateRec {        // one voxel step of one ray
  float sdf, w;
  uint32_t color, next;
};

__device__ __forceinline__ unsigned long long voxel_key(int x, int y, int z) {
  return (((unsigned long long)(x + kBias) & 0x1fffffull) << 42) | (((unsigned long long)(y + kBias) & 0x1fffffull) << 21) |
         ((unsigned long long)(z + kBias) & 0x1fffffull);
}

// updateTsdfVoxel's geometry (computeDistance + weight drop-off + sparsity compensation) [recalled]: the same
// operations as make_update in vgx_tsdf.hip
__device__ __forceinline__ void update_terms(float vs, const vgx_tsdf_config& c, float ox, float oy, float oz, float gx, float gy,
                                             float gz, int vx, int vy, int vz, float weight, float& sdf_out, float& w_out) {
  const float cx = ((float)vx + 0.5f) * vs, cy = ((float)vy + 0.5f) * vs, cz = ((float)vz + 0.5f) * vs;
  const float vvx = cx - ox, vvy = cy - oy, vvz = cz - oz;
  const float vpx = gx - ox, vpy = gy - oy, vpz = gz - oz;
  const float dist_G = norm3(vpx, vpy, vpz);
  const float dot = (vvx * vpx + vvy * vpy) + vvz * vpz;
  const float dist_G_V = dot / dist_G;
  const float sdf = dist_G - dist_G_V;
  float updated_weight = weight;
  const float trunc = c.default_truncation_distance;
  if (c.use_weight_dropoff && sdf < -vs) {
    updated_weight = weight * (trunc + sdf) / (trunc - vs);
    updated_weight = fmaxf(updated_weight, 0.0f);
  }
  if (c.use_sparsity_compensation_factor && fabsf(sdf) < trunc) updated_weight *= c.sparsity_compensation_factor;
  sdf_out = sdf;
  w_out = updated_weight;
}

// ---- the event log of the TRACE instantiations (csrc/bench/vgx_tsdf_diag.hip; layout: include/voxgraph_amd_bench.h) ----
// I.trace[0] = next free word (starts at kTraceHeaderWords), I.trace[1] = events that did not fit.  An event is a run of
// 64-bit words whose first word holds the kind in its low byte.
constexpr unsigned long long kTraceHeaderWords = 2;
enum : unsigned long long { kEvStartExchange = 1, kEvStartSkipped = 2, kEvRay = 3, kEvObservedExchange = 4, kEvFold = 5 };
__device__ __forceinline__ unsigned long long* trace_reserve(const TsdfIntegratorDev& I, unsigned long long words) {
  const unsigned long long at = atomicAdd(&I.trace[0], words);
  if (at + words > I.trace_words) {
    atomicAdd(&I.trace[1], 1ull);
    return nullptr;
  }
  return I.trace + at;
}
__device__ __forceinline__ void trace4(const TsdfIntegratorDev& I, unsigned long long a, unsigned long long b, unsigned long long c,
                                       unsigned long long d) {
  unsigned long long* w = trace_reserve(I, 4);
  if (w) {
    w[0] = a; w[1] = b; w[2] = c; w[3] = d;
  }
}

}  // namespace

template <bool STATS, bool TRACE>
__global__ __launch_bounds__(256) void tsdf_integrate_coop_kernel(TsdfLayerDev L, TsdfIntegratorDev I, float qw, float qx, float qy,
                                                                 float qz, float tx, float ty, float tz,
                                                                 const float* __restrict__ points_C,
                                                                 const uint32_t* __restrict__ rgba, long long n,
                                                                 int freespace_points, int cloud_width, int ablate) {
  __shared__ RayRec rays[256];
  __shared__ UpdateRec recs[kMaxRecs];
  __shared__ unsigned long long tkey[kTable];
  __shared__ uint32_t thead[kTable];
  __shared__ uint16_t occ[kMaxRecs];
  __shared__ uint16_t ray_list[2][256];   // the rays of the current pass / handed on to the next
  __shared__ uint32_t sh_n_rays, sh_next_ray, sh_n_recs, sh_n_occ, sh_n_next;
  __shared__ uint32_t ray_rounds_s[STATS ? 256 : 1];   // counted scans only: rounds a handed-on ray has spent so far
  // TRACE only: which point a queued ray belongs to and how many voxels its walk visits; which (point, step) a record is
  __shared__ uint32_t ray_point[TRACE ? 256 : 1];
  __shared__ unsigned long long ray_total[TRACE ? 256 : 1];
  __shared__ unsigned long long rec_id[TRACE ? kMaxRecs : 1];

  const vgx_tsdf_config& c = I.cfg;
  const int tid = (int)threadIdx.x, lane = tid & 63;
  unsigned long long st_updates = 0, st_dropped = 0, st_exch = 0, st_peeks = 0, st_blends = 0, st_voxels = 0, st_retries = 0,
                     st_overrun = 0, st_rounds_max = 0;

  const bool tracing = STATS && I.wg_stats != nullptr;
  if (tracing && tid == 0) I.wg_stats[(size_t)blockIdx.x * kWgStatWords + 0] = wall_clock64();
  for (int e = tid; e < kTable; e += 256) {
    tkey[e] = kEmptyKey;
    thead[e] = kNil;
  }
  if (tid == 0) {
    sh_n_rays = 0;
    sh_next_ray = 0;
    sh_n_recs = 0;
    sh_n_occ = 0;
    sh_n_next = 0;
  }
  ray_list[0][tid] = (uint16_t)tid;
  __syncthreads();

  // ---------------------------------------------------------------- phase 1: validity, start set, ray set-up
  {
    // Which point is this lane's?  An unorganised cloud: 256 consecutive points per workgroup.  An organised one
    // (cloud_width = points per row: sensor_msgs/PointCloud2.width): a tile of 16 x 16 beams, a wavefront = 4 rows of 16 --
    // the beams that end in one voxel are neighbours in BOTH directions, so a tile folds most of a voxel's updates in
    // its own LDS, where 256 consecutive beams of one ring share every voxel with the rings above and below, i.e. with
    // other workgroups (the per-voxel compare-and-swap chains of phase 3).  Which rays are cast and what they write is a
    // legal order either way.
    long long i = (long long)blockIdx.x * 256 + tid;
    bool have_point = i < n;
    if (cloud_width > 0) {
      const int tiles_x = (cloud_width + 15) >> 4;
      const int tile_y = (int)blockIdx.x / tiles_x, tile_x = (int)blockIdx.x - tile_y * tiles_x;
      const int col = tile_x * 16 + (tid & 15);
      const long long row = (long long)tile_y * 16 + (tid >> 4);
      i = row * cloud_width + col;
      have_point = col < cloud_width && i < n;
    }
    bool cast = false, valid = false, is_clearing = false;
    RayRec r;
    float pz = 0.0f, gx = 0.0f, gy = 0.0f, gz = 0.0f;
    unsigned long long v = 0ull;
    if (have_point) {
      const float px = points_C[3 * i], py = points_C[3 * i + 1];
      pz = points_C[3 * i + 2];
      // isPointValid
      valid = true;
      const float ray_distance = norm3(px, py, pz);
      if (ray_distance < c.min_ray_length_m) {
        valid = false;
      } else if (ray_distance > c.max_ray_length_m) {
        if (c.allow_clear || freespace_points) is_clearing = true; else valid = false;
      } else {
        is_clearing = freespace_points != 0;
      }
      transform_point(qw, qx, qy, qz, tx, ty, tz, px, py, pz, gx, gy, gz);
      const float sub_inv = c.start_voxel_subsampling_factor * L.voxel_size_inv;
      const int sx = grid_index(gx * sub_inv + 1e-6f), sy = grid_index(gy * sub_inv + 1e-6f), sz = grid_index(gz * sub_inv + 1e-6f);
      const unsigned int h = (unsigned int)sx + (unsigned int)sy * 17191u + (unsigned int)sz * 295530481u;
      v = (unsigned long long)h + I.start_offset;
    }
    // A lane whose left neighbour holds the same value would find it present: it does not exchange.  EVERY lane of the
    // wavefront takes part in the shuffles -- a lane without a point (a ragged tile of an organised cloud, the tail of an
    // unorganised one) or with an invalid one holds the marker, which no value equals (h + offset < 2^33).
    const unsigned long long vkey = valid ? v : kEmptyKey;
    const unsigned int left_lo = (unsigned int)__shfl_up((int)(unsigned int)vkey, 1);
    const unsigned int left_hi = (unsigned int)__shfl_up((int)(unsigned int)(vkey >> 32), 1);
    const bool same_as_left = lane > 0 && (((unsigned long long)left_hi << 32) | left_lo) == vkey;
    unsigned int left_point = 0;
    if (TRACE) left_point = (unsigned int)__shfl_up((int)(unsigned int)i, 1);
    if (valid && !same_as_left) {
      const unsigned long long old = atomicExch(&I.start_set[v & kSetMask], v);
      cast = old != v;
      if (TRACE) trace4(I, kEvStartExchange, (unsigned long long)i, v, old);
    } else if (TRACE && valid) {
      trace4(I, kEvStartSkipped, (unsigned long long)i, v, (unsigned long long)left_point);
    }
    if (cast) {
      const RayDda d = ray_setup(c, L.voxel_size_inv, tx, ty, tz, gx, gy, gz, is_clearing, false);
      cast = !d.bad;
      r.curr[0] = d.curr[0]; r.curr[1] = d.curr[1]; r.curr[2] = d.curr[2];
      r.sign_carry = (uint32_t)((d.sign[0] + 1) | ((d.sign[1] + 1) << 2) | ((d.sign[2] + 1) << 4));   // carry 0
      r.t_next[0] = d.t_next[0]; r.t_next[1] = d.t_next[1]; r.t_next[2] = d.t_next[2];
      r.t_step[0] = d.t_step[0]; r.t_step[1] = d.t_step[1]; r.t_step[2] = d.t_step[2];
      r.left_lo = (uint32_t)(unsigned long long)(d.steps + 1);   // the walk visits steps + 1 voxels
      r.left_hi = (uint32_t)((unsigned long long)(d.steps + 1) >> 32);
      r.point = (uint32_t)i;
      if (TRACE) trace4(I, kEvRay | (d.bad ? 0x100ull : 0ull), (unsigned long long)i, (unsigned long long)(d.steps + 1), 0ull);
    }
    const unsigned long long m = __ballot(cast);
    uint32_t base = 0;
    if (lane == 0 && m) base = atomicAdd(&sh_n_rays, (uint32_t)__popcll(m));
    base = (uint32_t)__shfl((int)base, 0);
    if (cast) {
      const uint32_t slot = base + (uint32_t)__popcll(m & ((1ull << lane) - 1ull));
      rays[slot] = r;
      if (STATS) ray_rounds_s[slot] = 0u;
      if (TRACE) {
        ray_point[slot] = (uint32_t)i;
        ray_total[slot] = ((unsigned long long)r.left_hi << 32) | r.left_lo;
      }
    }
  }
  __syncthreads();
  const uint32_t n_rays = sh_n_rays;
  if (tracing && tid == 0) {
    I.wg_stats[(size_t)blockIdx.x * kWgStatWords + 1] = wall_clock64();
    I.wg_stats[(size_t)blockIdx.x * kWgStatWords + 2] = 0ull;
  }

  // ---------------------------------------------------------------- phases 2 + 3: walk in rounds, flush
  // Lanes per ray, chosen per PASS from the rays the pass starts with: eight while they all fit side by side (32 groups),
  // four / two / one for more.  A pass with fewer than eight lanes per ray is short: after a few rounds a ray that is
  // still walking is handed on, with its state, to the next pass -- most rays stop within their first three voxels, the
  // few that walk on (a clear line of sight: up to max_ray_length / voxel_size steps) then get eight lanes and peeks
  // instead of crawling one exchange per round while 250 lanes idle (the config-2 city scans: 60 % of the points cast a
  // ray, the longest walks 110 voxels; 188 us per scan before this, profiles/r05_tsdf_racing.txt).
  uint32_t n_list = n_rays;
  int cur_list = 0;
  int lpr = 8, j = 0, gb = 0, budget = 1 << 30;
  uint32_t wmask = 0xffu;
  auto choose_lanes = [&](uint32_t rays_in_pass) {
    lpr = rays_in_pass <= 32u ? 8 : (rays_in_pass <= 64u ? 4 : (rays_in_pass <= 128u ? 2 : 1));
    j = lane & (lpr - 1);                 // this lane's step within the group's window
    gb = lane & ~(lpr - 1);               // first lane of the group
    wmask = (1u << lpr) - 1u;
    budget = lpr == 8 ? (1 << 30) : (lpr == 4 ? 2 : (lpr == 2 ? 3 : 4));   // rounds a ray gets in this pass
  };
  choose_lanes(n_list);
  // (a run of 2^24 observed voxels does not exist: the clamp only keeps mc + 1 - carry inside an int)
  const int mc = c.max_consecutive_ray_collisions < (1 << 24) ? c.max_consecutive_ray_collisions : (1 << 24);
  const int shift = L.vps_shift, vmask = L.vps - 1;
  const size_t vps3 = (size_t)L.vps * L.vps * L.vps;
  const float trunc = c.default_truncation_distance;

  // the group's ray (replicated over its lanes)
  int ray = -1, ray_rounds = 0;
  int cur[3] = {0, 0, 0}, sg[3] = {0, 0, 0};
  float tn[3] = {0, 0, 0}, ts[3] = {0, 0, 0};
  long long steps_left = 0;   // voxels of the walk not yet exchanged (the walk visits steps + 1 voxels)
  int carry = 0;
  uint32_t peekbits = 0;
  int nvalid = 0;
  float rgx = 0, rgy = 0, rgz = 0, rweight = 0;
  uint32_t rcolor = 0;
  unsigned long long rounds = 0;
  unsigned long long rpoint = 0, rtotal = 0;   // TRACE: the ray's point and the voxels its walk visits

  // Rounds are a wavefront's own business: its groups fetch rays, exchange, decide and queue records without waiting
  // for the other three (a round is ~600 instructions and a dozen LDS trips around ONE memory round trip, and with one
  // wavefront per SIMD nothing hides any of it: a barrier per round made every wavefront pay for the slowest).  The
  // workgroup meets only to fold: when every wavefront has run out of rays, or when the record pool may not take
  // another round of all four (each checks the count BEFORE a round and adds at most 64 records: 512 + 4 x 64 fit).
  bool all_done = n_rays == 0 || ablate >= 2;  // (ablate: attribution runs only, profiles/probes/run_racing_probe.sh)
  uint32_t wg_rounds = 0, wg_folds = 0, my_retry_max = 0;
  while (!all_done) {
    while (true) {
      const uint32_t pending = *(volatile uint32_t*)&sh_n_recs;
      // (A) a group without a ray takes the next one
      if (ray < 0 && *(volatile uint32_t*)&sh_next_ray < n_list) {
        uint32_t k = 0;
        if (j == 0) k = atomicAdd(&sh_next_ray, 1u);
        k = (uint32_t)__shfl((int)k, gb);
        if (k < n_list) {
          ray = (int)ray_list[cur_list][k];
          const RayRec& q = rays[ray];
          cur[0] = q.curr[0]; cur[1] = q.curr[1]; cur[2] = q.curr[2];
          sg[0] = (int)(q.sign_carry & 3u) - 1; sg[1] = (int)((q.sign_carry >> 2) & 3u) - 1; sg[2] = (int)((q.sign_carry >> 4) & 3u) - 1;
          tn[0] = q.t_next[0]; tn[1] = q.t_next[1]; tn[2] = q.t_next[2];
          ts[0] = q.t_step[0]; ts[1] = q.t_step[1]; ts[2] = q.t_step[2];
          steps_left = (long long)(((unsigned long long)q.left_hi << 32) | q.left_lo);
          carry = (int)(q.sign_carry >> 8);
          rounds = STATS ? (unsigned long long)ray_rounds_s[ray] : 0ull;
          peekbits = 0;
          nvalid = 0;
          {
            // the ray's end point, weight and colour again from its point (the loads are in flight beside the round's
            // exchanges: they are first needed for the update terms)
            const size_t pi = (size_t)q.point;
            const float px = points_C[3 * pi], py = points_C[3 * pi + 1], pz_ = points_C[3 * pi + 2];
            transform_point(qw, qx, qy, qz, tx, ty, tz, px, py, pz_, rgx, rgy, rgz);
            rweight = 1.0f;  // getVoxelWeight
            if (!c.use_const_weight) {
              const float dist_z = fabsf(pz_);
              rweight = dist_z > 1e-6f ? 1.0f / (dist_z * dist_z) : 0.0f;
            }
            rcolor = rgba ? rgba[pi] : 0u;
          }
          ray_rounds = 0;
          if (TRACE) {
            rpoint = ray_point[ray];
            rtotal = ray_total[ray];
          }
        }
      }
      if (!__any(ray >= 0) || pending > (uint32_t)kFlushAt) break;
      ++wg_rounds;
      // (B) one round of the group's ray
      bool emit = false;
      int vx = 0, vy = 0, vz = 0;
      float sdf = 0.0f, uw = 0.0f;
      unsigned long long rpoint_rec = 0, step_rec = 0;   // TRACE: this lane's record, should its step survive
      if (ray >= 0) {
        const int remaining = steps_left < (long long)lpr ? (int)steps_left : lpr;
        // exchanges that happen whatever they return: the stop needs a run of more than mc
        int must = mc + 1 - carry;
        must = must < 1 ? 1 : must;
        must = must > remaining ? remaining : must;
        int w = must;
        if (nvalid > must) {  // extend over the peeked prefix up to (and including) the step the peeks say stops the ray
          int run = carry, cut = -1;
          for (int k = 0; k < nvalid; ++k) {
            run = ((peekbits >> k) & 1u) ? run + 1 : 0;
            if (run > mc) {
              cut = k;
              break;
            }
          }
          w = cut >= 0 ? (cut + 1 > must ? cut + 1 : must) : nvalid;
          w = w > remaining ? remaining : w;
        }
        // this lane's voxel: the ray's state advanced j times (RayCaster::nextRayIndex, the same additions in the same order)
        int c0 = cur[0], c1 = cur[1], c2 = cur[2];
        float t0 = tn[0], t1 = tn[1], t2 = tn[2];
#pragma unroll
        for (int k = 0; k < kLanesPerRay; ++k) {
          if (k < lpr) {  // (uniform)
            if (k == j) {
              vx = c0; vy = c1; vz = c2;
            }
            if (k <= j) {  // (lane j also takes step j: its state afterwards is the ray's at pos + j + 1)
              int mm = 0;
              float tm = t0;
              if (t1 < tm) { mm = 1; tm = t1; }
              if (t2 < tm) { mm = 2; }
              c0 += mm == 0 ? sg[0] : 0; c1 += mm == 1 ? sg[1] : 0; c2 += mm == 2 ? sg[2] : 0;
              t0 += mm == 0 ? ts[0] : 0.0f; t1 += mm == 1 ? ts[1] : 0.0f; t2 += mm == 2 ? ts[2] : 0.0f;
            }
          }
        }
        const bool in_window = j < remaining;
        const bool do_x = j < w, do_peek = in_window && !do_x;
        const unsigned int h = (unsigned int)vx + (unsigned int)vy * 17191u + (unsigned int)vz * 295530481u;
        const unsigned long long v = (unsigned long long)h + I.observed_offset;
        unsigned long long got = 0ull;
        if (do_x) got = atomicExch(&I.observed_set[v & kSetMask], v);
        else if (do_peek) got = __hip_atomic_load(&I.observed_set[v & kSetMask], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        // (while that is in flight: what this step would write, should it survive)
        if (do_x) update_terms(L.voxel_size, c, tx, ty, tz, rgx, rgy, rgz, vx, vy, vz, rweight, sdf, uw);
        const bool seen = (do_x || do_peek) && got == v;
        // TRACE: every exchange with the step of the walk it belongs to (a peek decides nothing and is not an event)
        const unsigned long long step_id = TRACE ? (rtotal - (unsigned long long)steps_left) + (unsigned long long)j : 0ull;
        if (TRACE && do_x) trace4(I, kEvObservedExchange | (step_id << 8), rpoint, v, got);
        rpoint_rec = rpoint;
        step_rec = step_id;
        const uint32_t xbits = (uint32_t)(__ballot(do_x && seen) >> gb) & wmask;
        const uint32_t pbits = (uint32_t)(__ballot(do_peek && seen) >> gb) & wmask;
        if (STATS) {
          st_exch += do_x ? 1u : 0u;
          st_peeks += do_peek ? 1u : 0u;
        }
        ++rounds;
        // what the exchanges say
        int run = carry, cut = -1;
        for (int k = 0; k < w; ++k) {
          run = ((xbits >> k) & 1u) ? run + 1 : 0;
          if (run > mc) {
            cut = k;
            break;
          }
        }
        const int n_upd = cut >= 0 ? cut : w;
        if (STATS && cut >= 0 && j == 0) st_overrun += (unsigned)(w - 1 - cut);  // exchanges behind the stop (a peek went stale)
        emit = j < n_upd;
        const bool finished = cut >= 0 || (long long)w == steps_left;
        if (finished) {
          if (STATS && j == 0) st_rounds_max = rounds > st_rounds_max ? rounds : st_rounds_max;
          ray = -1;
        } else {
          carry = run;
          steps_left -= w;
          // the ray's state at pos + w: lane w - 1's state after its own step
          const int src = gb + w - 1;
          cur[0] = __shfl(c0, src); cur[1] = __shfl(c1, src); cur[2] = __shfl(c2, src);
          tn[0] = __shfl(t0, src); tn[1] = __shfl(t1, src); tn[2] = __shfl(t2, src);
          peekbits = pbits >> w;     // peeks of window positions w .. remaining - 1 become positions 0 ..
          nvalid = remaining - w;
          if (++ray_rounds >= budget) {  // still walking at the end of a short pass: on to the next one, state and all
            if (j == 0) {
              RayRec& q = rays[ray];
              q.curr[0] = cur[0]; q.curr[1] = cur[1]; q.curr[2] = cur[2];
              q.t_next[0] = tn[0]; q.t_next[1] = tn[1]; q.t_next[2] = tn[2];
              q.left_lo = (uint32_t)(unsigned long long)steps_left;
              q.left_hi = (uint32_t)((unsigned long long)steps_left >> 32);
              q.sign_carry = (q.sign_carry & 0xffu) | ((uint32_t)carry << 8);
              if (STATS) ray_rounds_s[ray] = (uint32_t)rounds;
              ray_list[cur_list ^ 1][atomicAdd(&sh_n_next, 1u)] = (uint16_t)ray;
            }
            ray = -1;
          }
        }
      }
      // (C) the round's surviving steps become records chained to their voxel
      {
        bool do_rec = false;
        if (emit) {
          do_rec = (unsigned)(vx + kBias) < (2u << 20) && (unsigned)(vy + kBias) < (2u << 20) && (unsigned)(vz + kBias) < (2u << 20);
          if (!do_rec) ++st_dropped;  // beyond +-2^20 voxels: no block table reaches there
        }
        // one LDS atomic per wavefront for the records' slots
        const unsigned long long rm = __ballot(do_rec);
        uint32_t rec_base = 0;
        if (rm) {
          if (lane == (int)__ffsll((long long)rm) - 1) rec_base = atomicAdd(&sh_n_recs, (uint32_t)__popcll(rm));
          rec_base = (uint32_t)__shfl((int)rec_base, (int)__ffsll((long long)rm) - 1);
        }
        uint32_t slot = 0;
        bool is_new = false;
        if (do_rec) {
          const unsigned long long key = voxel_key(vx, vy, vz);
          slot = (uint32_t)((key * 0x9E3779B97F4A7C15ull) >> 40) & (kTable - 1);
          while (true) {
            const unsigned long long prev = atomicCAS(&tkey[slot], kEmptyKey, key);
            if (prev == kEmptyKey) {
              is_new = true;
              break;
            }
            if (prev == key) break;
            slot = (slot + 1) & (kTable - 1);
          }
        }
        // the voxels this round meets for the first time, listed in lane order = ray by ray, step by step: phase 3
        // allocates new blocks in list order, so a scan of ONE ray allocates them in the order a sequential walk does
        const unsigned long long nm = __ballot(is_new);
        uint32_t occ_base = 0;
        if (nm) {
          if (lane == (int)__ffsll((long long)nm) - 1) occ_base = atomicAdd(&sh_n_occ, (uint32_t)__popcll(nm));
          occ_base = (uint32_t)__shfl((int)occ_base, (int)__ffsll((long long)nm) - 1);
        }
        if (is_new) occ[occ_base + (uint32_t)__popcll(nm & ((1ull << lane) - 1ull))] = (uint16_t)slot;
        if (do_rec) {
          const uint32_t rec = rec_base + (uint32_t)__popcll(rm & ((1ull << lane) - 1ull));
          UpdateRec u;
          u.sdf = sdf; u.w = uw; u.color = rcolor;
          u.next = atomicExch(&thead[slot], rec);
          recs[rec] = u;
          if (TRACE) rec_id[rec] = rpoint_rec | (step_rec << 32);
        }
      }
    }
    // (D) the workgroup meets: is this pass's list handed out and are its rays finished or handed on?  then the next pass
    const bool busy = ray >= 0 || sh_next_ray < n_list;
    const bool pass_done = __syncthreads_or(busy ? 1 : 0) == 0;
    const uint32_t handed_on = sh_n_next, pending_now = sh_n_recs;
    all_done = pass_done && handed_on == 0;
    if (pass_done && !all_done) {
      __syncthreads();  // (everybody has read the two counters)
      if (tid == 0) {
        sh_next_ray = 0;
        sh_n_next = 0;
      }
      cur_list ^= 1;
      n_list = handed_on;
      choose_lanes(n_list);
      __syncthreads();
    }
    if (all_done || pending_now > (uint32_t)kFlushAt) {
      if (tracing && tid == 0 && all_done) I.wg_stats[(size_t)blockIdx.x * kWgStatWords + 2] = wall_clock64();
      // ------------------------------------------------------------ phase 3: one lane per distinct voxel
      const uint32_t n_occ = sh_n_occ;
      wg_folds += n_occ;
      if (ablate >= 1)
        for (uint32_t o = (uint32_t)tid; o < n_occ; o += 256) {
          tkey[occ[o]] = kEmptyKey;
          thead[occ[o]] = kNil;
        }
      // New blocks first, in list order (one lane; a plain look at the table tells which voxels need one -- a handful per
      // scan once the layer exists): the order a sequential walk allocates them in when the scan is a single ray, and
      // SOME serial order otherwise, as under voxblox's block mutex.
      {
        bool need = false;
        for (uint32_t o = (uint32_t)tid; o < n_occ && ablate < 1; o += 256) {
          const unsigned long long key = tkey[occ[o] & 0x3ffu];
          const int bx = ((int)((key >> 42) & 0x1fffffull) - kBias) >> shift, by = ((int)((key >> 21) & 0x1fffffull) - kBias) >> shift,
                    bz = ((int)(key & 0x1fffffull) - kBias) >> shift;
          const int rx = bx - L.lut_min[0], ry = by - L.lut_min[1], rz = bz - L.lut_min[2];
          int e = -1;
          if ((unsigned)rx < (unsigned)L.lut_dim[0] && (unsigned)ry < (unsigned)L.lut_dim[1] && (unsigned)rz < (unsigned)L.lut_dim[2])
            e = L.lut[rx + L.lut_dim[0] * (ry + L.lut_dim[1] * rz)];
          if (e < 0) {
            occ[o] |= 0x8000u;
            need = true;
          }
        }
        if (__syncthreads_or(need ? 1 : 0)) {
          if (tid == 0)
            for (uint32_t o = 0; o < n_occ; ++o)
              if (occ[o] & 0x8000u) {
                const unsigned long long key = tkey[occ[o] & 0x3ffu];
                (void)get_or_allocate_block(L, ((int)((key >> 42) & 0x1fffffull) - kBias) >> shift,
                                            ((int)((key >> 21) & 0x1fffffull) - kBias) >> shift,
                                            ((int)(key & 0x1fffffull) - kBias) >> shift);
              }
          __syncthreads();
        }
      }
      // A lane owns up to kFoldsPerLane voxels of the table (n_occ <= kMaxRecs = 3 x 256).  Their memory operations go
      // out TOGETHER, stage by stage -- the two loads of all of them, then the compare-and-swaps of all of them, then the
      // colour writes -- so a flush is three round trips whatever the number of voxels, not three per voxel.
      constexpr int kFoldsPerLane = (kMaxRecs + 255) / 256;
      uint32_t f_head[kFoldsPerLane], f_oc[kFoldsPerLane], f_col[kFoldsPerLane], f_retries[kFoldsPerLane];
      unsigned long long f_old[kFoldsPerLane], f_want[kFoldsPerLane], f_prev[kFoldsPerLane];
      size_t f_at[kFoldsPerLane];
      bool f_on[kFoldsPerLane], f_blend[kFoldsPerLane];
      // TRACE: the voxel of a fold, whether it reached a block at all, whether its colour went out
      unsigned long long f_key[kFoldsPerLane];
      bool f_live[kFoldsPerLane], f_cw[kFoldsPerLane];
      // updateTsdfVoxel for every record of a chain in turn, on registers: {distance, weight} and colour after the chain
      auto fold = [&](uint32_t head, unsigned long long old, uint32_t oc, unsigned long long* want, uint32_t* col_out, bool* blends) {
        float d = __uint_as_float((unsigned)(old & 0xffffffffull)), W = __uint_as_float((unsigned)(old >> 32));
        uint32_t col = oc;
        bool any = false, any_blend = false;
        for (uint32_t q = head; q != kNil; q = recs[q].next) {
          const UpdateRec u = recs[q];
          const float new_weight = W + u.w;
          if (new_weight < 1e-6f) continue;  // kFloatEpsilon: the voxel is left alone
          const float new_sdf = (u.sdf * u.w + d * W) / new_weight;
          d = (new_sdf > 0.0f) ? fminf(trunc, new_sdf) : fmaxf(-trunc, new_sdf);
          if (fabsf(u.sdf) < trunc) {
            col = blended_color(col, u.color, W, u.w);
            any_blend = true;
          }
          W = fminf(c.max_weight, new_weight);
          any = true;
        }
        *want = pack_voxel(d, W);
        *col_out = col;
        *blends = any_blend;
        return any;
      };
      // stage 0: which voxel, which block (a plain, cacheable look at the table first -- an entry >= 0 never changes inside
      // a kernel; anything else -- free, being allocated, stale in this XCD's L2 -- goes through the atomic path)
#pragma unroll
      for (int e = 0; e < kFoldsPerLane; ++e) {
        const uint32_t o = (uint32_t)tid + 256u * (uint32_t)e;
        f_on[e] = o < n_occ && ablate < 1;
        f_head[e] = kNil;
        f_at[e] = 0;
        f_retries[e] = 0;
        f_key[e] = 0ull;
        f_live[e] = false;
        f_cw[e] = false;
        if (!f_on[e]) continue;
        const uint32_t slot = occ[o] & 0x3ffu;
        const unsigned long long key = tkey[slot];
        if (TRACE) f_key[e] = key;
        f_head[e] = thead[slot];
        tkey[slot] = kEmptyKey;
        thead[slot] = kNil;
        const int kx = (int)((key >> 42) & 0x1fffffull) - kBias, ky = (int)((key >> 21) & 0x1fffffull) - kBias,
                  kz = (int)(key & 0x1fffffull) - kBias;
        int bslot = -1;
        {
          const int bx = kx >> shift, by = ky >> shift, bz = kz >> shift;
          const int rx = bx - L.lut_min[0], ry = by - L.lut_min[1], rz = bz - L.lut_min[2];
          if ((unsigned)rx < (unsigned)L.lut_dim[0] && (unsigned)ry < (unsigned)L.lut_dim[1] && (unsigned)rz < (unsigned)L.lut_dim[2])
            bslot = L.lut[rx + L.lut_dim[0] * (ry + L.lut_dim[1] * rz)];
          if (bslot < 0) bslot = get_or_allocate_block(L, bx, by, bz);
        }
        uint32_t chain = 0;
        for (uint32_t q = f_head[e]; q != kNil; q = recs[q].next) ++chain;
        if (bslot < 0) {
          st_dropped += chain;
          f_on[e] = false;
          continue;
        }
        st_updates += chain;
        if (TRACE) f_live[e] = true;
        f_at[e] = (size_t)bslot * vps3 + (size_t)((kx & vmask) + L.vps * ((ky & vmask) + L.vps * (kz & vmask)));
      }
      // stage 1: the loads
#pragma unroll
      for (int e = 0; e < kFoldsPerLane; ++e) {
        f_old[e] = 0ull;
        f_oc[e] = 0u;
        if (f_on[e]) {
          f_old[e] = __hip_atomic_load(&L.voxels[f_at[e]], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          f_oc[e] = __hip_atomic_load(&L.rgba[f_at[e]], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
      }
      // stage 2: fold and publish, all of the lane's voxels together ...
      bool f_done[kFoldsPerLane];
#pragma unroll
      for (int e = 0; e < kFoldsPerLane; ++e) {
        f_done[e] = !f_on[e];
        f_want[e] = 0ull;
        f_col[e] = f_oc[e];
        f_blend[e] = false;
        if (f_on[e] && !fold(f_head[e], f_old[e], f_oc[e], &f_want[e], &f_col[e], &f_blend[e])) {
          f_on[e] = false;  // every record of the chain left the voxel alone
          f_done[e] = true;
          if (STATS) ++st_voxels;
        }
      }
#pragma unroll
      for (int e = 0; e < kFoldsPerLane; ++e) {
        f_prev[e] = f_old[e];
        if (!f_done[e]) f_prev[e] = atomicCAS(&L.voxels[f_at[e]], f_old[e], f_want[e]);
      }
      // ... then those that found another workgroup's update in between, each on its own: folded again over what it left
#pragma unroll
      for (int e = 0; e < kFoldsPerLane; ++e) {
        if (f_done[e]) continue;
        while (f_prev[e] != f_old[e]) {
          f_old[e] = f_prev[e];
          ++f_retries[e];
          if (STATS) ++st_retries;
          if (!fold(f_head[e], f_old[e], f_oc[e], &f_want[e], &f_col[e], &f_blend[e])) {
            f_on[e] = false;
            break;
          }
          f_prev[e] = atomicCAS(&L.voxels[f_at[e]], f_old[e], f_want[e]);
        }
      }
      // stage 3: colours.  A blend that leaves the colour as it was found needs no write (a scan without colours into a
      // layer without colours: every fold): this update's colour is then ordered at its LOAD -- the colour word is its own
      // atomic in any case, see the header comment -- and a third of a fold's round trips goes.
      uint32_t f_prevc[kFoldsPerLane];
#pragma unroll
      for (int e = 0; e < kFoldsPerLane; ++e) {
        f_prevc[e] = f_oc[e];
        if (!f_on[e]) continue;
        if (STATS) {
          ++st_voxels;
          if (f_blend[e])
            for (uint32_t q = f_head[e]; q != kNil; q = recs[q].next) st_blends += fabsf(recs[q].sdf) < trunc ? 1u : 0u;
        }
        if (f_blend[e] && f_col[e] != f_oc[e]) {
          f_prevc[e] = atomicCAS(&L.rgba[f_at[e]], f_oc[e], f_col[e]);
          if (TRACE) f_cw[e] = true;
        }
      }
#pragma unroll
      for (int e = 0; e < kFoldsPerLane; ++e) {
        if (!f_on[e]) continue;
        while (f_prevc[e] != f_oc[e]) {  // blend again over the colour found, with the weights this fold saw
          f_oc[e] = f_prevc[e];
          float W2 = __uint_as_float((unsigned)(f_old[e] >> 32));
          uint32_t col = f_oc[e];
          for (uint32_t q = f_head[e]; q != kNil; q = recs[q].next) {
            const UpdateRec u = recs[q];
            const float new_weight = W2 + u.w;
            if (new_weight < 1e-6f) continue;
            if (fabsf(u.sdf) < trunc) col = blended_color(col, u.color, W2, u.w);
            W2 = fminf(c.max_weight, new_weight);
          }
          ++f_retries[e];
          if (STATS) ++st_retries;
          if (TRACE) f_col[e] = col;
          if (col == f_oc[e]) {
            if (TRACE) f_cw[e] = false;
            break;
          }
          f_prevc[e] = atomicCAS(&L.rgba[f_at[e]], f_oc[e], col);
        }
        my_retry_max = f_retries[e] > my_retry_max ? f_retries[e] : my_retry_max;
      }
      if (TRACE) {
        // One event per fold that reached a block: the voxel, the word it folded over and the word it published (the same
        // word twice when every record left the voxel alone: nothing was published), the colour it blended over and the
        // colour it left, and its records in chain order -- the order updateTsdfVoxel was applied in.
#pragma unroll
        for (int e = 0; e < kFoldsPerLane; ++e) {
          if (!f_live[e]) continue;
          unsigned long long chain = 0;
          for (uint32_t q = f_head[e]; q != kNil; q = recs[q].next) ++chain;
          unsigned long long* w = trace_reserve(I, 6 + chain);
          if (!w) continue;
          const unsigned long long flags = (f_on[e] ? 1ull : 0ull) | (f_cw[e] ? 2ull : 0ull) | (f_blend[e] ? 4ull : 0ull);
          w[0] = kEvFold | (chain << 8) | (flags << 40);
          w[1] = f_key[e];
          w[2] = f_old[e];
          w[3] = f_on[e] ? f_want[e] : f_old[e];
          w[4] = (unsigned long long)f_oc[e] | ((unsigned long long)(f_on[e] ? f_col[e] : f_oc[e]) << 32);
          w[5] = (unsigned long long)f_at[e];
          unsigned long long k = 6;
          for (uint32_t q = f_head[e]; q != kNil; q = recs[q].next) w[k++] = rec_id[q];
        }
      }
      __syncthreads();
      if (tid == 0) {
        sh_n_recs = 0;
        sh_n_occ = 0;
      }
      __syncthreads();
    }
  }

  // Nothing is counted on an uncounted scan (every wavefront adding to ONE word at the end of a 300 000-point scan is
  // 4800 same-address atomics: tens of microseconds by themselves); dropped updates -- the GPU ran out of memory --
  // are reported in both modes.
  if (__any(st_dropped != 0ull)) {
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) st_dropped += __shfl_xor(st_dropped, off, 64);
    if (lane == 0) atomicAdd(L.dropped, st_dropped);
  }
  if (STATS) {
    // a counted scan: one row per workgroup, summed by reduce_wg_stats_kernel (no two workgroups share a word)
    __shared__ unsigned long long sh_stat[8];
    __shared__ uint32_t sh_retry_max;
    if (tid < 8) sh_stat[tid] = 0ull;
    if (tid == 0) sh_retry_max = 0;
    __syncthreads();
    if (st_updates) atomicAdd(&sh_stat[0], st_updates);
    if (st_rounds_max) atomicMax(&sh_stat[1], st_rounds_max);
    if (st_exch) atomicAdd(&sh_stat[2], st_exch);
    if (st_blends) atomicAdd(&sh_stat[3], st_blends);
    if (st_peeks) atomicAdd(&sh_stat[4], st_peeks);
    if (st_voxels) atomicAdd(&sh_stat[5], st_voxels);
    if (st_retries) atomicAdd(&sh_stat[6], st_retries);
    if (st_overrun) atomicAdd(&sh_stat[7], st_overrun);
    if (my_retry_max) atomicMax(&sh_retry_max, my_retry_max);
    __syncthreads();
    if (tracing && tid < 8) I.wg_stats[(size_t)blockIdx.x * kWgStatWords + 8 + tid] = sh_stat[tid];
    if (tracing && tid == 0) {
      unsigned long long* t = I.wg_stats + (size_t)blockIdx.x * kWgStatWords;
      t[3] = wall_clock64();
      t[4] = n_rays; t[5] = wg_rounds; t[6] = wg_folds; t[7] = sh_retry_max;
    }
  }
}

// counted scans: the workgroups' rows (words 8..15: updates, longest chain of rounds, exchanges, colour blends, peeks,
// per-voxel folds, repeated folds, exchanges behind a stop) -> I.n_updates[0..7]
static __global__ __launch_bounds__(256) void reduce_wg_stats_kernel(const unsigned long long* __restrict__ rows, long long n_rows,
                                                             unsigned long long* __restrict__ out) {
  __shared__ unsigned long long sh[8];
  if (threadIdx.x < 8) sh[threadIdx.x] = 0ull;
  __syncthreads();
  unsigned long long acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  for (long long r = threadIdx.x; r < n_rows; r += 256) {
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      const unsigned long long v = rows[r * kWgStatWords + 8 + k];
      acc[k] = k == 1 ? (v > acc[k] ? v : acc[k]) : acc[k] + v;
    }
  }
#pragma unroll
  for (int k = 0; k < 8; ++k) {
    if (k == 1) atomicMax(&sh[k], acc[k]); else atomicAdd(&sh[k], acc[k]);
  }
  __syncthreads();
  if (threadIdx.x < 8) out[threadIdx.x] = sh[threadIdx.x];
}

// The racing scan's launch; stats: the scan is a counted one (n_updates != NULL).  TRACE = false: launch_racing_scan of the
// product library (vgx_tsdf_coop.hip); TRACE = true: its event-logging twin (csrc/bench/vgx_tsdf_diag.hip; I.trace set).
template <bool TRACE>
inline hipError_t launch_racing_scan_t(hipStream_t stream, const TsdfLayerDev& L, const TsdfIntegratorDev& I, const float T[7],
                                       const float* d_points, const uint32_t* d_rgba, long long n, int freespace, bool stats,
                                       int cloud_width) {
  // an organised cloud (cloud_width points per row, n a whole number of rows): one workgroup per 16 x 16 tile of beams
  const bool tiled = cloud_width > 0 && n % cloud_width == 0;
  const long long wgs = racing_scan_workgroups(n, tiled ? cloud_width : 0);
  const dim3 grid((unsigned)wgs), block(256);
  const int cw = tiled ? cloud_width : 0;
  static const int ablate = getenv("VGX_TSDF_ABLATE") ? atoi(getenv("VGX_TSDF_ABLATE")) : 0;
  if (stats) {
    hipLaunchKernelGGL((tsdf_integrate_coop_kernel<true, TRACE>), grid, block, 0, stream, L, I, T[0], T[1], T[2], T[3], T[4], T[5],
                       T[6], d_points, d_rgba, n, freespace, cw, ablate);
    hipLaunchKernelGGL(reduce_wg_stats_kernel, dim3(1), block, 0, stream, I.wg_stats, (long long)grid.x, I.n_updates);
  } else
    hipLaunchKernelGGL((tsdf_integrate_coop_kernel<false, TRACE>), grid, block, 0, stream, L, I, T[0], T[1], T[2], T[3], T[4], T[5],
                       T[6], d_points, d_rgba, n, freespace, cw, ablate);
  return hipGetLastError();
}

}  // namespace vgx

#endif  // VGX_TSDF_COOP_KERNEL_H_
