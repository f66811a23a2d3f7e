// Internal host/device structures of libvoxgraph_amd.so (gfx950 only).
#ifndef VGX_INTERNAL_H_
#define VGX_INTERNAL_H_

#include <hip/hip_runtime.h>

#include <atomic>
#include <cstdint>
#include <mutex>
#include <condition_variable>
#include <random>
#include <string>
#include <vector>

#include "voxgraph_amd.h"

namespace vgx {

// ---------------------------------------------------------------------------
// Device-visible descriptors (plain structs, copied by value)
// ---------------------------------------------------------------------------

// A finished layer re-laid for sampling: one "apron brick" per voxblox block,
// (vps+1)^3 floats, x fastest.  Cell (i,j,k) of brick b holds voxel (i,j,k) of
// block b for i,j,k < vps and the first voxel plane of the +x/+y/+z neighbour
// blocks for index == vps, so the 8 trilinear neighbours of any base voxel of
// block b live in ONE brick.  Invalid voxels (unobserved / zero weight /
// missing neighbour block) are stored as NaN: validity travels with the value.
// Brick layouts.  A context chooses one for the submaps it will hold (vgx_ctx_set_brick_layout):
//   0  apron brick (default): (vps+1)^3 floats, x fastest.  A neighbourhood is four 8-byte x-pairs in four
//      rows (y, z), (y, z+1), (y+1, z), (y+1, z+1): 68 B, 1156 B apart -> four cache lines.  1.2 x memory.
//      Fastest for the all-points passes (HBM / VALU bound: fewest bytes).
//   1  quad brick: for every x in [0, vps], y, z in [0, vps) the float4 {d(x,y,z), d(x,y+1,z), d(x,y,z+1),
//      d(x,y+1,z+1)}, x fastest.  A neighbourhood is two consecutive float4: 32 contiguous bytes, one or
//      two cache lines.  4.25 x memory.  Fastest where evaluations are scattered (sampling mode: every
//      touched line is an HBM fetch): -14 % per evaluation of the shipped configuration, +18-25 % on the
//      all-points passes (profiles/ab_layout.sh).
//   2  4^3 sub-tiles with their own aprons (5^3 floats, padded to 128): the four x-pairs of a
//      neighbourhood lie within 128 B.  2 x memory.  Measured in between; kept as an experiment
//      (compile-time default only: -DVGX_BRICK_LAYOUT_DEFAULT=2).
#ifndef VGX_BRICK_LAYOUT_DEFAULT
#define VGX_BRICK_LAYOUT_DEFAULT 0
#endif
template <int VPS, int LAYOUT>
struct BrickLayout {
  static constexpr int B = VPS + 1;
  static constexpr int S = VPS / 4;
  static constexpr int cells = LAYOUT == 0 ? B * B * B : (LAYOUT == 1 ? B * VPS * VPS * 4 : S * S * S * 128);
  // float offset of a base voxel's neighbourhood anchor inside its brick
  __host__ __device__ static int anchor(int vx, int vy, int vz) {
    if (LAYOUT == 0) return vx + B * (vy + B * vz);
    if (LAYOUT == 1) return 4 * (vx + B * (vy + VPS * vz));
    return 128 * ((vx >> 2) + S * ((vy >> 2) + S * (vz >> 2))) + (vx & 3) + 5 * ((vy & 3) + 5 * (vz & 3));
  }
  // brick float index -> apron cell (cx, cy, cz in [0, VPS]); false: padding
  __host__ __device__ static bool decode(int i, int& cx, int& cy, int& cz) {
    if (LAYOUT == 0) {
      cx = i % B; cy = (i / B) % B; cz = i / (B * B);
      return true;
    }
    if (LAYOUT == 1) {
      const int comp = i & 3, e = i >> 2;
      cx = e % B; cy = (e / B) % VPS + (comp & 1); cz = e / (B * VPS) + (comp >> 1);
      return true;
    }
    const int s = i >> 7, k = i & 127;
    if (k >= 125) return false;
    cx = 4 * (s % S) + k % 5; cy = 4 * ((s / S) % S) + (k / 5) % 5; cz = 4 * (s / (S * S)) + k / 25;
    return true;
  }
};
inline size_t brick_cells(int vps, int layout) {
  if (vps == 16) return layout == 0 ? BrickLayout<16, 0>::cells : (layout == 1 ? BrickLayout<16, 1>::cells : BrickLayout<16, 2>::cells);
  return layout == 0 ? BrickLayout<8, 0>::cells : (layout == 1 ? BrickLayout<8, 1>::cells : BrickLayout<8, 2>::cells);
}

struct GridDev {
  const float* bricks;   // [n_blocks][BrickLayout<vps, layout>::cells]
  const int32_t* lut;    // dense block lookup [dim.z][dim.y][dim.x] -> brick or -1
  int32_t lut_min[3];
  int32_t lut_dim[3];
  float voxel_size, voxel_size_inv, block_size, block_size_inv;
  int32_t layout;        // brick layout of `bricks` (0 apron, 1 quad, 2 sub-tiles)
};

// Per-evaluation pose data of one constraint, computed on the host in the
// reference's own f32 arithmetic (registration_cost_function.cpp:69-110).
struct alignas(16) PosePack {
  float qw, qz;            // T_reading__reference rotation (yaw-only: x = y = 0)
  float tx, ty, tz;        // ... translation
  float cos_e, sin_e;      // cos/sin(yaw_reading)                    (.cpp:91-92)
  float cos_emo, sin_emo;  // cos/sin(yaw_reading - yaw_reference)    (.cpp:94-96)
  float dxs, dxc;          // (xe-xo)*sin_e , (xe-xo)*cos_e           (.cpp:225-226)
  float dys, dyc;          // (ye-yo)*sin_e , (ye-yo)*cos_e
  float k1, k2;            // dxs - dyc , dxc + dys (lean fused kernel)
  float pad;
};
static_assert(sizeof(PosePack) == 64, "PosePack must stay one 64-byte line");

// Static description of one constraint for the kernels.
struct alignas(16) ConstraintDev {
  GridDev grid;             // reading submap's sampling grid
  const float4* xyzd;       // reference submap's points {x,y,z,distance}
  const float* weight;      // ... weights
  // sampling mode (else all null): per residual two raw std::mt19937 outputs; the kernel turns
  // them into the WeightedSampler draw itself (generate_canonical -> * total -> upper_bound on the
  // cumulative weights -> optional Morton permutation), bit for bit what the host code did
  const uint32_t* sample_raw;
  const double* cumulative;   // [n_points] WeightedSampler::cumulative_item_weights_
  const int32_t* search_lut;  // [search_buckets + 1] first index of every draw bucket, or null
  int32_t search_buckets;     // K: a power of two
  const float4* sample_pts;   // batch: this evaluation's DRAWN POINTS {x,y,z,d}, by row (row0 + i); null: draw in place
  const int32_t* inv_order;   // uploaded index -> device index, or null
  int64_t n_points;
  const float4* chunk_bounds; // bounding spheres of consecutive kChunkPoints-point chunks
  int64_t n;                // num_residuals
  int64_t row0;             // first output row (stacked outputs)
  int64_t prow0;            // ... with every constraint padded to whole tiles: the blocked output layout (batch only)
  int32_t tile_points;      // batch: residuals per fused-pass tile of this constraint (all but its last tile)
  double factor;            // N / sum(w)                              (.cpp:274)
  double no_corr_cost;      // config.no_correspondence_cost
};

// A unit of work: `count` consecutive residuals of one constraint.
struct Tile {
  int32_t constraint;
  int32_t count;
  int64_t start;  // residual index within the constraint
};

// The weighted draw's std::upper_bound starts from a bucket table: with u = k / K the bound of
// u * total is precomputed, and upper_bound is monotone in its target, so the answer for any u in
// bucket k lies in [lut[k], lut[k + 1]].  K = the power of two >= n: the range holds about one
// element and the search becomes table load + four parallel loads, same result.
constexpr int kMinSearchBuckets = 4096;
constexpr int kMaxSearchBuckets = 1 << 24;
constexpr int kChunkPoints = 512;   // culling granule of the fused pass (= 256 threads x 2)
constexpr int kNormalSize = 45;     // per-constraint fused output (doubles)
constexpr int kPartialSize = 22;    // 21 unique products + reserved

// ---------------------------------------------------------------------------
// Host objects behind the opaque handles
// ---------------------------------------------------------------------------
struct Context {
  int device = 0;
  hipStream_t own_stream = nullptr;
  hipStream_t stream = nullptr;
  hipEvent_t ev_start = nullptr, ev_stop = nullptr;
  std::mutex mu;
  // The TSDF path (the ACTIVE submap: layers, integrators, scans) has its own stream and its own lock, so that a scan
  // neither queues behind a solver evaluation on `stream` nor waits for `mu` while another thread evaluates -- the
  // reference integrates on the ROS thread while optimizePoseGraph runs on a std::async thread (voxgraph_mapper.cpp:
  // 236-238), and what makes that safe is the same here: finished submaps are immutable (voxgraph_mapper.cpp:464-471).
  // The two sides meet in vgx_submap_from_tsdf_layer (finishSubmap: the TSDF side's stream is waited for by an event).
  // Lock order where both are taken: integrator -> tsdf_mu -> mu.
  hipStream_t tsdf_own_stream = nullptr;
  hipStream_t tsdf_stream = nullptr;
  bool stream_priorities = false;  // own streams: TSDF side at the device's highest priority, registration side at its lowest
  std::atomic<int> tsdf_integrators{0};  // TSDF integrators alive on this context: the fused pass leaves room for their scans
  hipEvent_t ev_tsdf_start = nullptr, ev_tsdf_stop = nullptr, ev_handover = nullptr;
  std::mutex tsdf_mu;
  std::mutex err_mu;       // guards last_error only (set_error is called with and without `mu`)
  std::string last_error;
  int cu_count = 256;
  int brick_layout = VGX_BRICK_LAYOUT_DEFAULT;  // of the submaps created from now on (vgx_ctx_set_brick_layout)
  int sampling_bricks = 1;  // VGX_SAMPLING_BRICKS_QUAD: all-sampling batches read quad bricks made on demand
  // Evaluation slots of the drop-in Evaluate path.  A call takes a free slot for its duration
  // (stream, ordering event, device staging for the f64 outputs, pinned + device staging for the
  // sampler's engine outputs -- all grown on demand and reused), so constructing a cost function
  // allocates nothing: voxgraph rebuilds every registration constraint before each solve
  // (pose_graph_interface.cpp:149-175).  Ceres uses 4 threads (pose_graph.cpp:96); with 8 slots a
  // caller practically never waits.
  struct EvalSlot {
    hipStream_t stream = nullptr;
    hipEvent_t order = nullptr;
    double* d_out = nullptr;
    int64_t out_rows = 0;
    uint32_t* d_raw = nullptr;
    uint32_t* h_raw = nullptr;
    int64_t raw_cap = 0;
    double* h_out = nullptr;   // pinned, kSmallOutputBytes: small evaluations come back in ONE copy
    bool busy = false;
  };
  static constexpr int kEvalSlots = 8;
  static constexpr size_t kSmallOutputBytes = 1u << 20;
  EvalSlot eval_slot[kEvalSlots];
  std::condition_variable slot_free;
};

// std::mt19937 (the engine of WeightedSampler, weighted_sampler.h:36-39) with its state in the
// open: the standard fixes the algorithm completely (seed 5489 -> 10000th output 4123659995), so
// this IS std::mt19937's stream, and the state can travel between the host and the device.
struct Mt19937 {
  static constexpr int kN = 624;
  uint32_t mt[kN];
  uint32_t idx = kN;  // next unread word of mt; kN = twist first (std::mt19937 after seed())
  Mt19937() { seed(5489u); }
  void seed(uint32_t s) {
    mt[0] = s;
    for (int i = 1; i < kN; ++i) mt[i] = 1812433253u * (mt[i - 1] ^ (mt[i - 1] >> 30)) + (uint32_t)i;
    idx = kN;
  }
  void twist() {
    for (int k = 0; k < kN; ++k) {
      uint32_t y = (mt[k] & 0x80000000u) | (mt[(k + 1) % kN] & 0x7fffffffu);
      mt[k] = mt[(k + 397) % kN] ^ (y >> 1) ^ ((y & 1u) ? 0x9908b0dfu : 0u);
    }
    idx = 0;
  }
  uint32_t operator()() {
    if (idx >= (uint32_t)kN) twist();
    uint32_t y = mt[idx++];
    y ^= y >> 11;
    y ^= (y << 7) & 0x9d2c5680u;
    y ^= (y << 15) & 0xefc60000u;
    y ^= y >> 18;
    return y;
  }
};
static_assert(sizeof(Mt19937) == 625 * 4, "Mt19937 is copied to the device as 625 words");

// A sampler engine that lives wherever it was advanced last: the drop-in Evaluate draws on the
// host (a few thousand outputs per call), the batched passes generate whole streams on the device
// (mt_generate_kernel).  Exactly one copy is current at any time.
struct SamplerEngine {
  Mt19937 host;
  uint32_t* d_state = nullptr;  // 625 words, allocated at first device use
  bool on_device = false;       // the device copy is the current one
};

struct PointSet {
  int64_t n = 0;
  uint64_t version = 0;  // bumped whenever the points are replaced (cost functions remember theirs)
  float4* d_xyzd = nullptr;
  float* d_weight = nullptr;
  float4* d_chunk_bounds = nullptr;  // per kChunkPoints points: bounding sphere {cx,cy,cz,r}
  float aabb_min[3] = {0, 0, 0}, aabb_max[3] = {0, 0, 0};  // of the point positions (n > 0)
  double sum_weight = 0;
  bool present = false;
  std::vector<int64_t> order;            // order[i] = uploaded index of point i (empty = identity)
  std::vector<int32_t> inv_order;        // uploaded index -> device index (for sampling)
  std::vector<double> cumulative_weight; // WeightedSampler::cumulative_item_weights_ (upload order)
  double* d_cumulative = nullptr;        // device copy, made when a sampling cost function is created
  int32_t search_buckets = 0;
  int32_t* d_search_lut = nullptr;       // bucket table of the draw (search_buckets + 1 entries), same moment
  int32_t* d_inv_order = nullptr;        // device copy of inv_order (Morton-sorted sets only)
  // WeightedSampler's mutable engine (weighted_sampler.h:36-39): ONE default-seeded std::mt19937 per
  // point set, shared by every cost function that samples this set (vgx_reg_config.sampler_seed == 0)
  SamplerEngine rng;
};

struct Grid {
  float* d_bricks = nullptr;
  bool present = false;
  int layout = 0;  // the context's brick layout when this grid was built
  // the same grid as quad bricks, made on demand from the apron bricks for a batch whose constraints all
  // SAMPLE (scattered evaluations: a neighbourhood in 32 contiguous bytes; vgx_ctx_set_sampling_bricks)
  float* d_quad = nullptr;
};

}  // namespace vgx

struct vgx_ctx_s : vgx::Context {};

struct vgx_submap_s {
  vgx_ctx ctx = nullptr;
  int32_t id = 0;
  float voxel_size = 0, voxel_size_inv = 0, block_size = 0, block_size_inv = 0;
  int32_t vps = 0;
  int32_t n_blocks = 0;
  std::vector<int32_t> block_index;  // host copy [n][3]
  int32_t* d_block_index = nullptr;
  int32_t* d_lut = nullptr;
  int32_t lut_min[3] = {0, 0, 0};
  int32_t lut_dim[3] = {0, 0, 0};
  // raw layers (voxblox layout), kept for point extraction
  float* d_tsdf_distance = nullptr;
  float* d_tsdf_weight = nullptr;
  float* d_esdf_distance = nullptr;
  uint8_t* d_esdf_observed = nullptr;
  vgx::Grid grid[2];       // [0] TSDF, [1] ESDF sampling grids
  vgx::PointSet points[2]; // by VGX_POINTS_*
  std::vector<int32_t> isosurface_blocks;  // block slots holding isosurface vertices (VSM:237-240)
  int32_t* d_iso_block_index = nullptr;    // [isosurface_blocks.size()][3]
  vgx::GridDev grid_dev(int which) const;
  int ensure_quad_grid(int which);  // apron bricks -> quad bricks, once (vgx_context.hip)
  // Lifetime (guarded by vgx::lifetime_mu()): cost functions made from this submap.  vgx_submap_destroy while users > 0
  // only records the request; the last vgx_reg_destroy carries it out (the reference's cost function holds
  // VoxgraphSubmap::ConstPtr: registration_cost_function.h -- a submap outlives every cost function built on it).
  int users = 0;
  bool destroy_requested = false;
};

struct vgx_reg_s {
  vgx_ctx ctx = nullptr;
  vgx_submap reference = nullptr;
  vgx_submap reading = nullptr;
  vgx_reg_config cfg{};
  int64_t num_residuals = 0;
  uint64_t points_version = 0;  // PointSet::version of the reference points at construction
  // sampling mode state: a private engine when cfg.sampler_seed != 0, else the point set's
  vgx::SamplerEngine rng;
  // vgx_reg_evaluate_device_f32 in sampling mode only (it does not wait for its kernel, so it
  // cannot borrow a slot): own staging for the engine outputs, allocated at first use
  uint32_t* d_sample_raw = nullptr;
  uint32_t* h_sample_raw = nullptr;
  // Drop-in Evaluate calls arrive from several Ceres threads (pose_graph.cpp:96), each on its own
  // cost function; each call borrows one of the context's evaluation slots (Context::EvalSlot).
  std::mutex mu;                // two threads on the SAME cost function are serialised
  int draw_raw(uint32_t* out);  // 2 engine outputs per residual, drawn on the host
  vgx::SamplerEngine& engine(); // the stream this cost function samples from
  bool sampling() const { return cfg.sampling_ratio != -1.0f; }
  bool points_current() const;  // the reference submap still holds the points this was built on
  vgx::ConstraintDev describe() const;
  // Lifetime (vgx::lifetime_mu()): batches that list this cost function; vgx_reg_destroy while users > 0 is deferred to
  // the last vgx_reg_batch_destroy (ceres::Problem owns its cost functions for as long as it evaluates them).
  int users = 0;
  bool destroy_requested = false;
};

struct vgx_reg_batch_s {
  vgx_ctx ctx = nullptr;
  int32_t n = 0;
  int32_t n_global = 0;
  int layout = 0;                     // brick layout of every reading grid in the batch
  std::vector<vgx_reg> regs;
  std::vector<int32_t> node_pair;     // [n][2]
  std::vector<int32_t> global_index;  // [n]
  std::vector<int64_t> row_offset;    // [n+1]
  std::vector<vgx::Tile> tiles;
  vgx::ConstraintDev* d_desc = nullptr;
  vgx::PosePack* d_pack = nullptr;
  vgx::PosePack* h_pack = nullptr;    // pinned, 2 x n (double-buffered staging)
  hipEvent_t pack_copied[2] = {nullptr, nullptr};  // H2D of staging half k finished
  int pack_turn = 0;
  vgx::Tile* d_tiles = nullptr;
  vgx::Tile* d_draw_tiles = nullptr;     // the sampling constraints' tiles in the draw kernel's launch order
  int32_t n_draw_tiles = 0;
  float4* d_drawn = nullptr;             // sampling: the point {x,y,z,d} every row uses in this evaluation (reg_gather_points_kernel)
  int32_t* d_drawn_idx = nullptr;        // ... and its index in the point set (reg_draw_kernel)
  unsigned char* d_tile_dead = nullptr;  // per materialising-pass tile, per launch: every chunk culled (rows are zeros)
  int32_t* d_tile_first = nullptr;    // [n+1] first tile of each constraint
  double* d_partials = nullptr;       // [n_tiles][4 wavefronts][kPartialSize]
  double* d_normal = nullptr;         // [n][45] (internal, when caller passes none)
  double* h_normal = nullptr;         // pinned [n][45]: staging of the host copies (normal_host / cost_host)
  int32_t* d_node_pair = nullptr;
  int32_t* d_global_index = nullptr;
  // fused pass: coarser tiles, and node -> incident (constraint<<1 | side) CSR
  std::vector<vgx::Tile> reduce_tiles;  // constraint-major; the device copy is re-ordered for launch at
                                        // the first evaluation (XCD-aware, make_xcd_order)
  std::vector<vgx::ConstraintDev> host_desc;
  std::vector<int32_t> host_tile_first;
  bool launch_order_made = false;
  bool launch_order_grouped = false, points_order_grouped = false;  // what make_xcd_order decided
  bool points_order_made = false;              // same, for the materialising pass's 1024-point tiles
  std::vector<int32_t> host_points_tile_first;
  vgx::Tile* d_reduce_tiles = nullptr;
  int32_t csr_nodes = 0;
  int32_t* d_node_first = nullptr;
  int32_t* d_node_items = nullptr;
  // sampling constraints (sampling_ratio != -1): one stream job per distinct engine, in order of
  // first appearance; the engine's constraints consume consecutive ranges of its stream in
  // constraint order, as successive Evaluate calls on one thread would (RCF:113-122)
  struct StreamJob {
    vgx::SamplerEngine* engine;
    int64_t offset;   // first word in d_raw
    int64_t count;    // words generated per evaluation
  };
  std::vector<StreamJob> stream_jobs;
  uint32_t* d_raw = nullptr;       // all engine outputs of one evaluation
  void* d_stream_jobs = nullptr;   // device copy of {state*, out*, count} per job
  bool any_sampling = false;
  bool holds_regs = false;         // regs[*]->users were incremented (vgx_reg_batch_create succeeded)
  // vgx_reg_batch_evaluate_rows_f64 / _fetch_rows_f64: f64 rows the batch keeps itself (allocated at first use), and -- while
  // they are small enough (kRowsMirrorLimit) -- their pinned host mirror, filled by ONE device-to-host copy per evaluation
  double* d_rows[3] = {nullptr, nullptr, nullptr};   // residuals [R], jac_ref [R][4], jac_read [R][4]
  char* h_rows = nullptr;                            // [R x 8][R x 32][R x 32] pinned, or NULL (too large: fetches copy slices)
  bool rows_have[3] = {false, false, false};         // what the last rows evaluation produced
  bool rows_mirrored = false;                        // the mirror holds the last evaluation (its copy may still be in flight)
};

// ---------------------------------------------------------------------------
// helpers
// ---------------------------------------------------------------------------
namespace vgx {
#ifndef VGX_TOOLING_LIBRARY
int set_error(vgx_ctx ctx, int code, const std::string& msg);
#else
inline int set_error(vgx_ctx ctx, int code, const std::string& msg);
#endif

// the one lock behind the users / destroy_requested fields of submaps and cost functions (vgx_context.hip)
std::mutex& lifetime_mu();

// Scope-bound device scratch: freed on every exit path (the VGX_HIP macro returns early).
struct DeviceScratch {
  void* p = nullptr;
  DeviceScratch() = default;
  DeviceScratch(const DeviceScratch&) = delete;
  DeviceScratch& operator=(const DeviceScratch&) = delete;
  ~DeviceScratch() {
    if (p) (void)hipFree(p);
  }
  hipError_t alloc(size_t bytes) { return hipMalloc(&p, bytes); }
  template <typename T>
  T* as() const {
    return static_cast<T*>(p);
  }
};
void set_global_error(const std::string& msg);

#define VGX_HIP(ctx, call)                                                      \
  do {                                                                          \
    hipError_t e__ = (call);                                                    \
    if (e__ != hipSuccess)                                                      \
      return vgx::set_error((ctx), VGX_ERR_HIP,                                 \
                            std::string(#call) + ": " + hipGetErrorString(e__)); \
  } while (0)

// kernels' launch wrappers implemented in the .hip files
#ifndef VGX_TOOLING_LIBRARY
int launch_brickify(vgx_submap sm, int which);
int build_block_lut(vgx_submap sm);
#else
inline int launch_brickify(vgx_submap sm, int which);
inline int build_block_lut(vgx_submap sm);
#endif
int build_chunk_bounds(vgx_ctx ctx, PointSet& ps);
// frees a point set's device arrays and leaves it empty (present = false) with a new version
void reset_point_set(PointSet& ps);
// sampler engines: make the host / the device copy the current one (ctx->mu held)
int engine_to_host(vgx_ctx ctx, SamplerEngine& e);
int engine_to_device(vgx_ctx ctx, SamplerEngine& e);
void make_pose_pack(const double ref_pose[4], const double read_pose[4], PosePack* out);
std::vector<Tile> make_tiles(int32_t constraint, int64_t n, int tile_points);
constexpr int kBlockThreads = 256;
constexpr int kPointsPerThread = 4;  // measured 5.09-5.28 / 5.06-5.12 / 5.77-5.79 ms at 2 / 4 / 8 (config 3)
constexpr int kTilePoints = kBlockThreads * kPointsPerThread;
}  // namespace vgx

// The library is built with -fvisibility=hidden: the C ABI of include/voxgraph_amd.h is what it exports -- plus these three
// hooks, C linkage like everything else, for libvoxgraph_amd_bench.so alone (csrc/bench/: scene generators, memory
// ceilings, the racing TSDF kernel's event log: test and benchmark tooling built from these same internal headers, kept
// out of the product library).  In no public header; vgx_context.hip defines them on vgx::set_error / launch_brickify /
// build_block_lut.
extern "C" {
VGX_API int vgx_internal_set_error(vgx_ctx ctx, int code, const char* msg);
VGX_API int vgx_internal_launch_brickify(vgx_submap sm, int which);
VGX_API int vgx_internal_build_block_lut(vgx_submap sm);
}
#ifdef VGX_TOOLING_LIBRARY
// inside the tooling library the internal names resolve to the hooks
namespace vgx {
inline int set_error(vgx_ctx ctx, int code, const std::string& msg) { return vgx_internal_set_error(ctx, code, msg.c_str()); }
inline int launch_brickify(vgx_submap sm, int which) { return vgx_internal_launch_brickify(sm, which); }
inline int build_block_lut(vgx_submap sm) { return vgx_internal_build_block_lut(sm); }
}  // namespace vgx
#endif

#endif
