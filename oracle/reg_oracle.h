/*
 * oracle/reg_oracle.h -- CPU restatement of voxgraph's SDF-to-SDF registration
 * cost (REG path).  TEST INFRASTRUCTURE ONLY.
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may
 * link or call this.  The product (libvoxgraph_amd.so) never does.
 *
 * PARITY STATUS: pinned to the reference's own source for everything voxgraph
 * owns; unpinned for what voxblox / minkindr own.
 *   - oracle/_ref/libref_reg.so is /root/reference's registration_cost_function.cpp
 *     (+ its weighted_sampler / registration_point headers) compiled, from where it
 *     lies, against the stand-in headers of oracle/ref_shims (Eigen, glog, minkindr,
 *     voxblox, Ceres, ROS are absent from this image).  This file reproduces its
 *     residuals and Jacobians value for value (tests/test_ref_pin.py, live and
 *     through the committed fixture tests/golden/ref_reg_config1.npz).
 *   - the two 3x4 pose-Jacobian matrices also match vectors generated from the
 *     reference's sympy derivation voxgraph/scripts/jacobians_xyz_yaw.py
 *     (tests/golden/jacobians_xyz_yaw.json, made by tests/golden/make_golden.py)
 *   - the mt19937 stream matches the C++ standard's known answer
 *   - interpolation + analytic Jacobians match closed forms (plane) and central
 *     differences.
 * PARITY UNPINNED for everything tagged [recalled]: the un-vendored dependencies
 * (voxblox grid + interpolator, minkindr transformation) are restated from
 * knowledge of their public sources, here AND in the shims the reference source is
 * compiled against; the reference ships no tests or fixtures that would pin them.
 */
#ifndef VOXGRAPH_AMD_ORACLE_REG_ORACLE_H_
#define VOXGRAPH_AMD_ORACLE_REG_ORACLE_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* A voxblox::Layer<VoxelType> reduced to what the REG path reads:
 * per-voxel distance and the interpolator's validity predicate
 * (EsdfVoxel.observed, or TsdfVoxel.weight > 0).  [recalled] */
typedef struct orc_layer {
  float voxel_size;      /* Layer::voxel_size_ */
  float voxel_size_inv;  /* 1.0 / voxel_size_ (f32) */
  float block_size;      /* voxels_per_side * voxel_size */
  float block_size_inv;  /* 1.0 / block_size (f32) */
  int vps;               /* voxels_per_side (voxblox default 16) */
  int n_blocks;
  const int32_t* block_index; /* [n_blocks][3] */
  const float* distance;      /* [n_blocks][vps^3], linear = x + vps*(y + vps*z) */
  const uint8_t* valid;       /* [n_blocks][vps^3] */
  /* block lookup: dense table over the block AABB (stands in for the
   * unordered_map<BlockIndex, Block::Ptr>) */
  int32_t lut_min[3];
  int32_t lut_dim[3];
  int32_t* lut;
} orc_layer;

int orc_layer_init(orc_layer* L, float voxel_size, int vps, int n_blocks,
                   const int32_t* block_index, const float* distance,
                   const uint8_t* valid);
void orc_layer_free(orc_layer* L);

/* voxblox::Interpolator<V>::getVoxelsAndQVector(pos, voxels, q) [recalled].
 * Returns 1 and fills dist8/q8 when all 8 neighbours exist and are valid. */
int orc_get_voxels_and_q(const orc_layer* L, const float pos[3],
                         float dist8[8], float q8[8]);

/* T_reading__reference = exp(reading)^-1 * exp(reference) in f32
 * (registration_cost_function.cpp:69-88,109-110). q = (w,x,y,z). */
void orc_relative_transform(const double ref_pose[4], const double read_pose[4],
                            float q_wxyz[4], float t[3]);

/* q (x) p (x) q^-1 + t, Eigen's _transformVector form [recalled]. */
void orc_transform_point(const float q_wxyz[4], const float t[3],
                         const float p[3], float out[3]);

/* The two 3x4 matrices of registration_cost_function.cpp:214-227, row-major. */
void orc_pose_jacobian_matrices(float xi, float yi, const double ref_pose[4],
                                const double read_pose[4], float M_ref[12],
                                float M_read[12]);

typedef struct orc_reg_config {
  double no_correspondence_cost; /* registration_cost_function.h:32 */
} orc_reg_config;

/* RegistrationCostFunction::Evaluate, deterministic mode
 * (registration_cost_function.cpp:58-298 with sampling_ratio == -1).
 * points: xyz[3n] (reference-submap frame), dist[n], weight[n].
 * jac_ref / jac_read: [n][4] row-major, either may be NULL; want_jac == 0
 * mirrors `jacobians == nullptr`.
 * If sample_idx != NULL the call mirrors sampling mode: point i is
 * points[sample_idx[i]] with weight forced to 1 (.cpp:118-122) and
 * n is the number of samples.
 * Returns 1 (true) or 0 (false: summed weight == 0, .cpp:273). */
int orc_reg_evaluate(const orc_layer* reading, const orc_reg_config* cfg,
                     int64_t n, const float* xyz, const float* dist,
                     const float* weight, const int64_t* sample_idx,
                     const double ref_pose[4], const double read_pose[4],
                     int want_jac, double* residuals, double* jac_ref,
                     double* jac_read);

/* Same evaluation, but emits what the fused GPU mode emits: the cost
 * sum(r^2), J^T r (8) and the upper triangle of J^T J (36, row-major over the
 * 8 stacked parameters [ref(4), read(4)]), accumulated in f64. */
int orc_reg_evaluate_normal(const orc_layer* reading, const orc_reg_config* cfg,
                            int64_t n, const float* xyz, const float* dist,
                            const float* weight, const double ref_pose[4],
                            const double read_pose[4], double* cost,
                            double Jtr[8], double JtJ[36]);

/* ---- sampling (weighted_sampler_inl.h:18-28) -------------------------- */
typedef struct orc_mt19937 {
  uint32_t mt[624];
  int idx;
} orc_mt19937;
void orc_mt19937_seed(orc_mt19937* g, uint32_t seed); /* default 5489 */
uint32_t orc_mt19937_next(orc_mt19937* g);
/* libstdc++ uniform_real_distribution<double>(0,1) over mt19937
 * (generate_canonical<double,53>: two 32-bit draws). */
double orc_uniform01(orc_mt19937* g);
/* WeightedSampler::getRandomItem's index: upper_bound on the cumulative
 * weights of random * total. cumulative[n] is built as addItem does
 * (weighted_sampler_inl.h:5-16). */
int64_t orc_weighted_draw(orc_mt19937* g, const double* cumulative, int64_t n);

/* ---- registration point extraction (voxgraph_submap.cpp:144-201) ------ */
/* Scans blocks in the given order and voxels by linear index; emits points
 * {voxel centre, distance (ESDF if esdf_distance != NULL else TSDF), TSDF
 * weight}.  Returns the number of points; out arrays may be NULL to count. */
int64_t orc_find_relevant_voxels(float voxel_size, int vps, int n_blocks,
                                 const int32_t* block_index,
                                 const float* tsdf_distance,
                                 const float* tsdf_weight,
                                 const float* esdf_distance,
                                 double min_voxel_weight,
                                 double max_voxel_distance, float* xyz,
                                 float* dist, float* weight);

#ifdef __cplusplus
}
#endif
#endif
