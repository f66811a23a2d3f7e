/*
 * oracle/reg_oracle.c -- CPU restatement of the REG path.  TEST INFRASTRUCTURE.
 * See reg_oracle.h for the usage rule and the parity status (pinned / unpinned parts).
 *
 * Build with -ffp-contract=off: the reference is compiled without FMA
 * contraction guarantees and its precision map is
 * restated operation by operation.
 *
 * Citations are relative to /root/reference/voxgraph/ :
 *   RCF = src/backend/constraint/cost_functions/registration_cost_function.cpp
 *   RCH = include/voxgraph/backend/constraint/cost_functions/registration_cost_function.h
 *   WSI = include/voxgraph/frontend/submap_collection/weighted_sampler_inl.h
 *   VSM = src/frontend/submap_collection/voxgraph_submap.cpp
 */
#include "reg_oracle.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>

/* ------------------------------------------------------------------------ */
/* voxblox Layer / Block geometry [recalled]                                 */
/* ------------------------------------------------------------------------ */

/* voxblox::kCoordinateEpsilon */
static const float kCoordinateEpsilon = 1e-6f;

int orc_layer_init(orc_layer* L, float voxel_size, int vps, int n_blocks,
                   const int32_t* block_index, const float* distance,
                   const uint8_t* valid) {
  memset(L, 0, sizeof(*L));
  L->voxel_size = voxel_size;
  L->voxel_size_inv = 1.0f / voxel_size;
  L->vps = vps;
  L->block_size = (float)vps * voxel_size;
  L->block_size_inv = 1.0f / L->block_size;
  L->n_blocks = n_blocks;
  L->block_index = block_index;
  L->distance = distance;
  L->valid = valid;
  if (n_blocks <= 0) {
    L->lut = NULL;
    return 0;
  }
  int32_t mn[3], mx[3];
  for (int a = 0; a < 3; ++a) mn[a] = mx[a] = block_index[a];
  for (int b = 1; b < n_blocks; ++b)
    for (int a = 0; a < 3; ++a) {
      int32_t v = block_index[3 * b + a];
      if (v < mn[a]) mn[a] = v;
      if (v > mx[a]) mx[a] = v;
    }
  size_t total = 1;
  for (int a = 0; a < 3; ++a) {
    L->lut_min[a] = mn[a];
    L->lut_dim[a] = mx[a] - mn[a] + 1;
    total *= (size_t)L->lut_dim[a];
  }
  L->lut = (int32_t*)malloc(total * sizeof(int32_t));
  if (!L->lut) return -1;
  for (size_t i = 0; i < total; ++i) L->lut[i] = -1;
  for (int b = 0; b < n_blocks; ++b) {
    size_t ix = (size_t)(block_index[3 * b + 0] - mn[0]);
    size_t iy = (size_t)(block_index[3 * b + 1] - mn[1]);
    size_t iz = (size_t)(block_index[3 * b + 2] - mn[2]);
    L->lut[ix + (size_t)L->lut_dim[0] * (iy + (size_t)L->lut_dim[1] * iz)] = b;
  }
  return 0;
}

void orc_layer_free(orc_layer* L) {
  free(L->lut);
  L->lut = NULL;
}

/* Layer::getBlockPtrByIndex: slot or -1 (nullptr). */
static int32_t block_slot(const orc_layer* L, const int32_t b[3]) {
  if (!L->lut) return -1;
  int32_t r[3];
  for (int a = 0; a < 3; ++a) {
    r[a] = b[a] - L->lut_min[a];
    if (r[a] < 0 || r[a] >= L->lut_dim[a]) return -1;
  }
  return L->lut[(size_t)r[0] +
                (size_t)L->lut_dim[0] *
                    ((size_t)r[1] + (size_t)L->lut_dim[1] * (size_t)r[2])];
}

/* voxblox::getGridIndexFromPoint: floor(p * inv + eps), f32. */
static int32_t grid_index(float p, float inv) {
  return (int32_t)floorf(p * inv + kCoordinateEpsilon);
}

/* Block::computeCoordinatesFromVoxelIndex along one axis:
 * origin + (float(idx) + 0.5) * voxel_size; the product is formed in double
 * ("0.5" literal) and rounded to f32 -- identical to the f32 product. */
static float voxel_centre(float origin, int32_t idx, float voxel_size) {
  float c = (float)(((double)(float)idx + 0.5) * (double)voxel_size);
  return origin + c;
}

/* Interpolator::setIndexes + getVoxelsAndQVector + getQVector [recalled]. */
int orc_get_voxels_and_q(const orc_layer* L, const float pos[3],
                         float dist8[8], float q8[8]) {
  const int vps = L->vps;
  int32_t block[3], vox[3];
  /* setIndexes: the block containing pos must exist */
  for (int a = 0; a < 3; ++a) block[a] = grid_index(pos[a], L->block_size_inv);
  if (block_slot(L, block) < 0) return 0;
  for (int a = 0; a < 3; ++a) {
    /* computeTruncatedVoxelIndexFromCoordinates */
    float origin = (float)block[a] * L->block_size;
    int32_t v = grid_index(pos[a] - origin, L->voxel_size_inv);
    if (v > vps - 1) v = vps - 1;
    if (v < 0) v = 0;
    /* shift to the bottom-left neighbour */
    float centre_offset = pos[a] - voxel_centre(origin, v, L->voxel_size);
    if (centre_offset < 0) {
      v--;
      if (v < 0) {
        block[a]--;
        v += vps;
      }
    }
    vox[a] = v;
  }
  /* getVoxelsAndQVector: neighbour k at base + (k>>2&1, k>>1&1, k&1) */
  for (int k = 0; k < 8; ++k) {
    if (block_slot(L, block) < 0) return 0;
    int32_t nb[3], nv[3];
    int off[3] = {(k >> 2) & 1, (k >> 1) & 1, k & 1};
    for (int a = 0; a < 3; ++a) {
      nb[a] = block[a];
      nv[a] = vox[a] + off[a];
      if (nv[a] >= vps) {
        nb[a]++;
        nv[a] -= vps;
      }
    }
    int32_t slot = block_slot(L, nb);
    if (slot < 0) return 0;
    if (k == 0) {
      /* getQVector with the bottom-left voxel's centre */
      float d[3];
      for (int a = 0; a < 3; ++a) {
        float origin = (float)nb[a] * L->block_size;
        d[a] = (pos[a] - voxel_centre(origin, nv[a], L->voxel_size)) *
               L->voxel_size_inv;
      }
      q8[0] = 1.0f;
      q8[1] = d[0];
      q8[2] = d[1];
      q8[3] = d[2];
      q8[4] = d[0] * d[1];
      q8[5] = d[1] * d[2];
      q8[6] = d[2] * d[0];
      q8[7] = d[0] * d[1] * d[2];
    }
    size_t lin = (size_t)nv[0] + (size_t)vps * ((size_t)nv[1] + (size_t)vps * (size_t)nv[2]);
    size_t at = (size_t)slot * (size_t)vps * vps * vps + lin;
    dist8[k] = L->distance[at];
    if (!L->valid[at]) return 0;
  }
  return 1;
}

/* ------------------------------------------------------------------------ */
/* minkindr QuatTransformationTemplate<float> [recalled]                     */
/* ------------------------------------------------------------------------ */

typedef struct {
  float w, x, y, z;
} quatf;

/* RotationQuaternionTemplate<float>::exp((0,0,psi)): internal math in double
 * (Grassia 1998 form), narrowed to float on construction. */
static quatf quat_exp_yaw(float psi) {
  float nrm = sqrtf(0.0f * 0.0f + 0.0f * 0.0f + psi * psi);
  double theta = (double)nrm;
  double na;
  if (theta < pow(2.220446049250313e-16, 0.25)) {
    na = 0.5 + (theta * theta) * (1.0 / 48.0);
  } else {
    na = sin(theta * 0.5) / theta;
  }
  double ct = cos(theta * 0.5);
  quatf q;
  q.w = (float)ct;
  q.x = (float)(0.0 * na);
  q.y = (float)(0.0 * na);
  q.z = (float)((double)psi * na);
  return q;
}

static quatf quat_conj(quatf q) {
  quatf r = {q.w, -q.x, -q.y, -q.z};
  return r;
}

/* Eigen generic quaternion product */
static quatf quat_mul(quatf a, quatf b) {
  quatf r;
  r.w = a.w * b.w - a.x * b.x - a.y * b.y - a.z * b.z;
  r.x = a.w * b.x + a.x * b.w + a.y * b.z - a.z * b.y;
  r.y = a.w * b.y + a.y * b.w + a.z * b.x - a.x * b.z;
  r.z = a.w * b.z + a.z * b.w + a.x * b.y - a.y * b.x;
  return r;
}

/* Eigen QuaternionBase::_transformVector: uv = 2 u x v; v + w uv + u x uv */
static void quat_rotate(quatf q, const float v[3], float out[3]) {
  float uv[3];
  uv[0] = q.y * v[2] - q.z * v[1];
  uv[1] = q.z * v[0] - q.x * v[2];
  uv[2] = q.x * v[1] - q.y * v[0];
  uv[0] += uv[0];
  uv[1] += uv[1];
  uv[2] += uv[2];
  float c[3];
  c[0] = q.y * uv[2] - q.z * uv[1];
  c[1] = q.z * uv[0] - q.x * uv[2];
  c[2] = q.x * uv[1] - q.y * uv[0];
  out[0] = v[0] + q.w * uv[0] + c[0];
  out[1] = v[1] + q.w * uv[1] + c[1];
  out[2] = v[2] + q.w * uv[2] + c[2];
}

void orc_transform_point(const float q_wxyz[4], const float t[3],
                         const float p[3], float out[3]) {
  quatf q = {q_wxyz[0], q_wxyz[1], q_wxyz[2], q_wxyz[3]};
  float r[3];
  quat_rotate(q, p, r);
  out[0] = r[0] + t[0];
  out[1] = r[1] + t[1];
  out[2] = r[2] + t[2];
}

/* RCF:69-88 (f64 parameters narrowed into a float Vector6, exp) and
 * RCF:109-110 (T_reading__reference = T_mission__reading.inverse() *
 * T_mission__reference). */
void orc_relative_transform(const double ref_pose[4], const double read_pose[4],
                            float q_wxyz[4], float t[3]) {
  float t_ref[3] = {(float)ref_pose[0], (float)ref_pose[1], (float)ref_pose[2]};
  float t_read[3] = {(float)read_pose[0], (float)read_pose[1],
                     (float)read_pose[2]};
  quatf q_ref = quat_exp_yaw((float)ref_pose[3]);
  quatf q_read = quat_exp_yaw((float)read_pose[3]);
  /* inverse(): (q^-1, -(q^-1 * t)) */
  quatf q_inv = quat_conj(q_read);
  float t_inv[3];
  quat_rotate(q_inv, t_read, t_inv);
  t_inv[0] = -t_inv[0];
  t_inv[1] = -t_inv[1];
  t_inv[2] = -t_inv[2];
  /* operator*: (q_a * q_b, t_a + q_a * t_b) */
  quatf q = quat_mul(q_inv, q_ref);
  float rt[3];
  quat_rotate(q_inv, t_ref, rt);
  q_wxyz[0] = q.w;
  q_wxyz[1] = q.x;
  q_wxyz[2] = q.y;
  q_wxyz[3] = q.z;
  t[0] = t_inv[0] + rt[0];
  t[1] = t_inv[1] + rt[1];
  t[2] = t_inv[2] + rt[2];
}

/* ------------------------------------------------------------------------ */
/* RegistrationCostFunction::Evaluate                                        */
/* ------------------------------------------------------------------------ */

/* RCF:91-100: trig and translations as float, std::cos(float) = cosf */
typedef struct {
  float cos_e, sin_e, cos_emo, sin_emo, xe, ye, xo, yo;
} pose_scalars;

static pose_scalars pose_scalars_from(const double ref_pose[4],
                                      const double read_pose[4]) {
  pose_scalars s;
  float yaw_ref = (float)ref_pose[3];
  float yaw_read = (float)read_pose[3];
  s.cos_e = cosf(yaw_read);
  s.sin_e = sinf(yaw_read);
  s.cos_emo = cosf(yaw_read - yaw_ref);
  s.sin_emo = sinf(yaw_read - yaw_ref);
  s.xe = (float)read_pose[0];
  s.ye = (float)read_pose[1];
  s.xo = (float)ref_pose[0];
  s.yo = (float)ref_pose[1];
  return s;
}

/* RCF:214-227 */
static void pose_matrices(const pose_scalars* s, float xi, float yi,
                          float Mo[12], float Me[12]) {
  Mo[0] = s->cos_e;  Mo[1] = s->sin_e; Mo[2] = 0;  Mo[3] = xi * s->sin_emo - yi * s->cos_emo;
  Mo[4] = -s->sin_e; Mo[5] = s->cos_e; Mo[6] = 0;  Mo[7] = xi * s->cos_emo + yi * s->sin_emo;
  Mo[8] = 0;         Mo[9] = 0;        Mo[10] = 1; Mo[11] = 0;
  Me[0] = -s->cos_e; Me[1] = -s->sin_e; Me[2] = 0;
  Me[3] = -xi * s->sin_emo + yi * s->cos_emo + (s->xe - s->xo) * s->sin_e - (s->ye - s->yo) * s->cos_e;
  Me[4] = s->sin_e;  Me[5] = -s->cos_e; Me[6] = 0;
  Me[7] = -xi * s->cos_emo - yi * s->sin_emo + (s->xe - s->xo) * s->cos_e + (s->ye - s->yo) * s->sin_e;
  Me[8] = 0;         Me[9] = 0;         Me[10] = -1; Me[11] = 0;
}

void orc_pose_jacobian_matrices(float xi, float yi, const double ref_pose[4],
                                const double read_pose[4], float M_ref[12],
                                float M_read[12]) {
  pose_scalars s = pose_scalars_from(ref_pose, read_pose);
  pose_matrices(&s, xi, yi, M_ref, M_read);
}

/* interp_table_ (RCH:73-81), B_1 of http://spie.org/samples/PM159.pdf */
static const float kInterpTable[8][8] = {
    {1, 0, 0, 0, 0, 0, 0, 0},   {-1, 0, 0, 0, 1, 0, 0, 0},
    {-1, 0, 1, 0, 0, 0, 0, 0},  {-1, 1, 0, 0, 0, 0, 0, 0},
    {1, 0, -1, 0, -1, 0, 1, 0}, {1, -1, -1, 1, 0, 0, 0, 0},
    {1, -1, 0, 0, -1, 1, 0, 0}, {-1, 1, 1, -1, 1, -1, -1, 1}};

/* One point of the hot loop, RCF:113-268, unscaled.  Returns interp_possible. */
static int eval_point(const orc_layer* reading, const orc_reg_config* cfg,
                      const float q[4], const float t[3],
                      const pose_scalars* s, const float p_ref[3], float d_ref,
                      float w, int want_jac, double* residual, float Jo[4],
                      float Je[4]) {
  float p_read[3];
  orc_transform_point(q, t, p_ref, p_read); /* RCF:128-129 */
  float distances[8], q_vector[8];
  int ok = orc_get_voxels_and_q(reading, p_read, distances, q_vector);
  /* c = interp_table_ * distances^T (f32) */
  float c[8];
  if (ok) {
    for (int r = 0; r < 8; ++r) {
      float acc = 0.0f;
      for (int k = 0; k < 8; ++k) acc += kInterpTable[r][k] * distances[k];
      c[r] = acc;
    }
    /* RCF:158-163: f32 dot widened to f64, residual arithmetic in f64 */
    float dot = 0.0f;
    for (int k = 0; k < 8; ++k) dot += q_vector[k] * c[k];
    double reading_distance = (double)dot;
    *residual = ((double)d_ref - reading_distance) * (double)w;
  } else {
    *residual = (double)w * cfg->no_correspondence_cost; /* RCF:165-166 */
  }
  if (!want_jac) return ok;
  if (ok) {
    /* RCF:183-202: inv, Dx, Dy, Dz are doubles; matrix entries round to f32 */
    double inv = (double)reading->voxel_size_inv;
    double Dx = (double)q_vector[1], Dy = (double)q_vector[2],
           Dz = (double)q_vector[3];
    float pQ[8][3] = {{0, 0, 0},
                      {(float)inv, 0, 0},
                      {0, (float)inv, 0},
                      {0, 0, (float)inv},
                      {(float)(inv * Dy), (float)(inv * Dx), 0},
                      {0, (float)(inv * Dz), (float)(inv * Dy)},
                      {(float)(inv * Dz), 0, (float)(inv * Dx)},
                      {(float)(inv * Dy * Dz), (float)(inv * Dx * Dz),
                       (float)(inv * Dx * Dy)}};
    /* RCF:204-205: (distances * interp_table_^T) * pQ_pr, f32 */
    float g[3];
    for (int a = 0; a < 3; ++a) {
      float acc = 0.0f;
      for (int k = 0; k < 8; ++k) acc += c[k] * pQ[k][a];
      g[a] = acc;
    }
    float Mo[12], Me[12];
    pose_matrices(s, p_ref[0], p_ref[1], Mo, Me);
    /* RCF:234-239: (-w * g) * M, f32 */
    float h[3] = {-w * g[0], -w * g[1], -w * g[2]};
    for (int col = 0; col < 4; ++col) {
      Jo[col] = h[0] * Mo[col] + h[1] * Mo[4 + col] + h[2] * Mo[8 + col];
      Je[col] = h[0] * Me[col] + h[1] * Me[4 + col] + h[2] * Me[8 + col];
    }
  } else {
    for (int col = 0; col < 4; ++col) Jo[col] = Je[col] = 0.0f; /* RCF:241-242 */
  }
  return ok;
}

int orc_reg_evaluate(const orc_layer* reading, const orc_reg_config* cfg,
                     int64_t n, const float* xyz, const float* dist,
                     const float* weight, const int64_t* sample_idx,
                     const double ref_pose[4], const double read_pose[4],
                     int want_jac, double* residuals, double* jac_ref,
                     double* jac_read) {
  float q[4], t[3];
  orc_relative_transform(ref_pose, read_pose, q, t);
  pose_scalars s = pose_scalars_from(ref_pose, read_pose);
  double summed_reference_weight = 0; /* RCF:60 */
  for (int64_t i = 0; i < n; ++i) {
    int64_t src = sample_idx ? sample_idx[i] : i;
    float w = sample_idx ? 1.0f : weight[src]; /* RCF:118-122 */
    summed_reference_weight += (double)w;     /* RCF:124 */
    float Jo[4], Je[4];
    eval_point(reading, cfg, q, t, &s, &xyz[3 * src], dist[src], w, want_jac,
               &residuals[i], Jo, Je);
    if (want_jac) {
      for (int col = 0; col < 4; ++col) {
        if (jac_ref) jac_ref[4 * i + col] = (double)Jo[col];   /* RCF:254-259 */
        if (jac_read) jac_read[4 * i + col] = (double)Je[col]; /* RCF:261-266 */
      }
    }
  }
  if (summed_reference_weight == 0) return 0; /* RCF:273 */
  double factor = (double)n / summed_reference_weight; /* RCF:274 */
  for (int64_t i = 0; i < n; ++i) {
    residuals[i] *= factor;
    if (want_jac) {
      for (int col = 0; col < 4; ++col) {
        if (jac_ref) jac_ref[4 * i + col] *= factor;
        if (jac_read) jac_read[4 * i + col] *= factor;
      }
    }
  }
  return 1;
}

int orc_reg_evaluate_normal(const orc_layer* reading, const orc_reg_config* cfg,
                            int64_t n, const float* xyz, const float* dist,
                            const float* weight, const double ref_pose[4],
                            const double read_pose[4], double* cost,
                            double Jtr[8], double JtJ[36]) {
  float q[4], t[3];
  orc_relative_transform(ref_pose, read_pose, q, t);
  pose_scalars s = pose_scalars_from(ref_pose, read_pose);
  double sw = 0, rr = 0;
  double g[8], H[36];
  memset(g, 0, sizeof(g));
  memset(H, 0, sizeof(H));
  for (int64_t i = 0; i < n; ++i) {
    float w = weight[i];
    sw += (double)w;
    double r;
    float Jo[4], Je[4];
    eval_point(reading, cfg, q, t, &s, &xyz[3 * i], dist[i], w, 1, &r, Jo, Je);
    double J[8];
    for (int c = 0; c < 4; ++c) {
      J[c] = (double)Jo[c];
      J[4 + c] = (double)Je[c];
    }
    rr += r * r;
    int k = 0;
    for (int a = 0; a < 8; ++a) {
      g[a] += J[a] * r;
      for (int b = a; b < 8; ++b) H[k++] += J[a] * J[b];
    }
  }
  if (sw == 0) return 0;
  double f = (double)n / sw;
  *cost = rr * f * f;
  for (int a = 0; a < 8; ++a) Jtr[a] = g[a] * f * f;
  for (int k = 0; k < 36; ++k) JtJ[k] = H[k] * f * f;
  return 1;
}

/* ------------------------------------------------------------------------ */
/* WeightedSampler (std::mt19937 + uniform_real_distribution<double>)        */
/* ------------------------------------------------------------------------ */

void orc_mt19937_seed(orc_mt19937* g, uint32_t seed) {
  g->mt[0] = seed;
  for (int i = 1; i < 624; ++i)
    g->mt[i] = 1812433253u * (g->mt[i - 1] ^ (g->mt[i - 1] >> 30)) + (uint32_t)i;
  g->idx = 624;
}

uint32_t orc_mt19937_next(orc_mt19937* g) {
  if (g->idx >= 624) {
    for (int i = 0; i < 624; ++i) {
      uint32_t y = (g->mt[i] & 0x80000000u) | (g->mt[(i + 1) % 624] & 0x7fffffffu);
      uint32_t v = g->mt[(i + 397) % 624] ^ (y >> 1);
      if (y & 1u) v ^= 0x9908b0dfu;
      g->mt[i] = v;
    }
    g->idx = 0;
  }
  uint32_t y = g->mt[g->idx++];
  y ^= (y >> 11);
  y ^= (y << 7) & 0x9d2c5680u;
  y ^= (y << 15) & 0xefc60000u;
  y ^= (y >> 18);
  return y;
}

/* libstdc++ generate_canonical<double, 53, mt19937>: k = 2 draws,
 * sum = lo + hi * 2^32, divided by 2^64; 1.0 is nudged below 1. */
double orc_uniform01(orc_mt19937* g) {
  const double range = 4294967296.0;
  double lo = (double)orc_mt19937_next(g);
  double hi = (double)orc_mt19937_next(g);
  double ret = (lo + hi * range) / (range * range);
  if (ret >= 1.0) ret = nextafter(1.0, 0.0);
  return ret;
}

/* WSI:18-28 */
int64_t orc_weighted_draw(orc_mt19937* g, const double* cumulative, int64_t n) {
  double random_number = orc_uniform01(g);
  double target = random_number * cumulative[n - 1];
  /* std::upper_bound: first element > target */
  int64_t lo = 0, hi = n;
  while (lo < hi) {
    int64_t mid = lo + (hi - lo) / 2;
    if (!(target < cumulative[mid])) lo = mid + 1; else hi = mid;
  }
  return lo;
}

/* ------------------------------------------------------------------------ */
/* VoxgraphSubmap::findRelevantVoxelIndices, VSM:144-201                     */
/* ------------------------------------------------------------------------ */
int64_t orc_find_relevant_voxels(float voxel_size, int vps, int n_blocks,
                                 const int32_t* block_index,
                                 const float* tsdf_distance,
                                 const float* tsdf_weight,
                                 const float* esdf_distance,
                                 double min_voxel_weight,
                                 double max_voxel_distance, float* xyz,
                                 float* dist, float* weight) {
  const size_t nvox = (size_t)vps * vps * vps;
  const float block_size = (float)vps * voxel_size;
  int64_t n = 0;
  for (int b = 0; b < n_blocks; ++b) {
    for (size_t lin = 0; lin < nvox; ++lin) {
      size_t at = (size_t)b * nvox + lin;
      float tw = tsdf_weight[at], td = tsdf_distance[at];
      /* VSM:177-178, compared in double */
      if ((double)tw > min_voxel_weight && (double)fabsf(td) < max_voxel_distance) {
        if (xyz) {
          int32_t v[3] = {(int32_t)(lin % (size_t)vps),
                          (int32_t)((lin / (size_t)vps) % (size_t)vps),
                          (int32_t)(lin / ((size_t)vps * vps))};
          for (int a = 0; a < 3; ++a) {
            float origin = (float)block_index[3 * b + a] * block_size;
            xyz[3 * n + a] = voxel_centre(origin, v[a], voxel_size);
          }
          dist[n] = esdf_distance ? esdf_distance[at] : td; /* VSM:185-192 */
          weight[n] = tw;                                   /* VSM:195-197 */
        }
        ++n;
      }
    }
  }
  return n;
}
