/*
 * oracle/tsdf_oracle.c -- see tsdf_oracle.h.  TEST INFRASTRUCTURE, parity unpinned,
 * everything [recalled] from voxblox (not vendored in /root/reference).
 * Build with -ffp-contract=off.
 */
#include "tsdf_oracle.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>

#include "tsdf_oracle_impl.h" /* the per-point / per-voxel steps, shared with the replay checker (tsdf_replay.c) */

void orc_tsdf_config_default(orc_tsdf_config* c) {
  c->default_truncation_distance = 0.1f;
  c->max_weight = 10000.0f;
  c->voxel_carving_enabled = 1;
  c->min_ray_length_m = 0.1f;
  c->max_ray_length_m = 5.0f;
  c->use_const_weight = 0;
  c->allow_clear = 1;
  c->use_weight_dropoff = 1;
  c->use_sparsity_compensation_factor = 0;
  c->sparsity_compensation_factor = 1.0f;
  c->start_voxel_subsampling_factor = 2.0f;
  c->max_consecutive_ray_collisions = 2;
  c->clear_checks_every_n_frames = 1;
  c->integration_order = 1;
  c->enable_anti_grazing = 0;
}

/* ---- Layer<TsdfVoxel>: hash map BlockIndex -> block ---------------------- */
struct orc_tsdf_layer {
  float voxel_size, voxel_size_inv, block_size, block_size_inv;
  int vps, vps_inv_shift;
  size_t nvox;
  int n_blocks, cap_blocks;
  int32_t* block_index; /* [cap][3] */
  float* distance;      /* [cap][nvox] */
  float* weight;
  uint8_t* rgba;        /* [cap][nvox][4] */
  /* open addressing table of slots */
  int32_t* table;
  size_t table_size;
};

static size_t block_hash(int32_t x, int32_t y, int32_t z) {
  /* voxblox AnyIndexHash: x + y*17191 + z*17191^2 */
  return (size_t)((int64_t)x + (int64_t)y * 17191 + (int64_t)z * 17191 * 17191);
}

orc_tsdf_layer* orc_tsdf_layer_create(float voxel_size, int vps) {
  orc_tsdf_layer* L = (orc_tsdf_layer*)calloc(1, sizeof(*L));
  L->voxel_size = voxel_size;
  L->voxel_size_inv = 1.0f / voxel_size;
  L->vps = vps;
  L->block_size = (float)vps * voxel_size;
  L->block_size_inv = 1.0f / L->block_size;
  L->nvox = (size_t)vps * vps * vps;
  L->table_size = 1024;
  L->table = (int32_t*)malloc(L->table_size * sizeof(int32_t));
  for (size_t i = 0; i < L->table_size; ++i) L->table[i] = -1;
  return L;
}

void orc_tsdf_layer_destroy(orc_tsdf_layer* L) {
  if (!L) return;
  free(L->block_index);
  free(L->distance);
  free(L->weight);
  free(L->rgba);
  free(L->table);
  free(L);
}

int orc_tsdf_layer_num_blocks(const orc_tsdf_layer* L) { return L->n_blocks; }

void orc_tsdf_layer_download(const orc_tsdf_layer* L, int32_t* block_index, float* distance,
                             float* weight, uint8_t* rgba) {
  size_t n = (size_t)L->n_blocks;
  if (block_index) memcpy(block_index, L->block_index, n * 3 * sizeof(int32_t));
  if (distance) memcpy(distance, L->distance, n * L->nvox * sizeof(float));
  if (weight) memcpy(weight, L->weight, n * L->nvox * sizeof(float));
  if (rgba) memcpy(rgba, L->rgba, n * L->nvox * 4);
}

static void table_insert(orc_tsdf_layer* L, int slot) {
  const int32_t* b = &L->block_index[3 * slot];
  size_t h = block_hash(b[0], b[1], b[2]) & (L->table_size - 1);
  while (L->table[h] >= 0) h = (h + 1) & (L->table_size - 1);
  L->table[h] = slot;
}

/* Layer::allocateBlockPtrByIndex: existing slot or a new zero-initialised block
 * (TsdfVoxel{distance 0, weight 0, color 0}) */
static int layer_get_or_allocate(orc_tsdf_layer* L, int32_t x, int32_t y, int32_t z) {
  size_t h = block_hash(x, y, z) & (L->table_size - 1);
  while (L->table[h] >= 0) {
    const int32_t* b = &L->block_index[3 * L->table[h]];
    if (b[0] == x && b[1] == y && b[2] == z) return L->table[h];
    h = (h + 1) & (L->table_size - 1);
  }
  if (L->n_blocks == L->cap_blocks) {
    int cap = L->cap_blocks ? 2 * L->cap_blocks : 64;
    L->block_index = (int32_t*)realloc(L->block_index, (size_t)cap * 3 * sizeof(int32_t));
    L->distance = (float*)realloc(L->distance, (size_t)cap * L->nvox * sizeof(float));
    L->weight = (float*)realloc(L->weight, (size_t)cap * L->nvox * sizeof(float));
    L->rgba = (uint8_t*)realloc(L->rgba, (size_t)cap * L->nvox * 4);
    L->cap_blocks = cap;
  }
  int slot = L->n_blocks++;
  L->block_index[3 * slot] = x;
  L->block_index[3 * slot + 1] = y;
  L->block_index[3 * slot + 2] = z;
  memset(&L->distance[(size_t)slot * L->nvox], 0, L->nvox * sizeof(float));
  memset(&L->weight[(size_t)slot * L->nvox], 0, L->nvox * sizeof(float));
  memset(&L->rgba[(size_t)slot * L->nvox * 4], 0, L->nvox * 4);
  if ((size_t)L->n_blocks * 2 > L->table_size) {
    L->table_size *= 2;
    L->table = (int32_t*)realloc(L->table, L->table_size * sizeof(int32_t));
    for (size_t i = 0; i < L->table_size; ++i) L->table[i] = -1;
    for (int s = 0; s < L->n_blocks; ++s) table_insert(L, s);
  } else {
    L->table[h] = slot;
  }
  return slot;
}

/* ---- ApproxHashSet<20, 10000, GlobalIndex, LongIndexHash> ---------------- */
#define ORC_SET_BITS 20
#define ORC_SET_SIZE (1u << ORC_SET_BITS)
#define ORC_SET_MASK (ORC_SET_SIZE - 1u)
#define ORC_FULL_RESET 10000u

typedef struct {
  uint64_t offset;
  uint64_t* slots;
} approx_set;

static void approx_set_init(approx_set* s) {
  s->offset = 0;
  s->slots = (uint64_t*)calloc(ORC_SET_SIZE, sizeof(uint64_t));
  /* the 0 hash would look present in every zeroed slot: poison its slot */
  s->slots[s->offset & ORC_SET_MASK] = UINT64_MAX;
}

static void approx_set_reset(approx_set* s) {
  if (++s->offset >= ORC_FULL_RESET) {
    memset(s->slots, 0, ORC_SET_SIZE * sizeof(uint64_t));
    s->offset = 0;
    s->slots[s->offset & ORC_SET_MASK] = UINT64_MAX;
  }
}

/* replaceHash: true if the slot did NOT already hold this hash */
static int approx_set_exchange(approx_set* s, const int64_t idx[3], uint64_t* value, uint64_t* returned) {
  uint64_t v = long_index_hash(idx) + s->offset;
  uint64_t* slot = &s->slots[v & ORC_SET_MASK];
  uint64_t old = *slot;
  *slot = v;
  *value = v;
  *returned = old;
  return old != v;
}

static int approx_set_replace(approx_set* s, const int64_t idx[3]) {
  uint64_t v = long_index_hash(idx) + s->offset;
  uint64_t* slot = &s->slots[v & ORC_SET_MASK];
  uint64_t old = *slot;
  *slot = v;
  return old != v;
}

static uint64_t pack_voxel_key(const int64_t v[3], int clearing);

/* ---- integrator ----------------------------------------------------------- */
struct orc_tsdf_integrator {
  orc_tsdf_config cfg;
  orc_tsdf_layer* layer;
  approx_set start_set, observed_set;
  /* optional event log of the scans to come, in the racing kernel's format (tsdf_replay.h; orc_tsdf_integrator_set_log) */
  uint64_t* log;
  int64_t log_cap, log_n, log_lost;
  int64_t reset_counter; /* function-static in voxblox */
};

orc_tsdf_integrator* orc_tsdf_integrator_create(const orc_tsdf_config* cfg, orc_tsdf_layer* layer) {
  orc_tsdf_integrator* I = (orc_tsdf_integrator*)calloc(1, sizeof(*I));
  I->cfg = *cfg;
  I->layer = layer;
  approx_set_init(&I->start_set);
  approx_set_init(&I->observed_set);
  return I;
}

void orc_tsdf_integrator_destroy(orc_tsdf_integrator* I) {
  if (!I) return;
  free(I->start_set.slots);
  free(I->observed_set.slots);
  free(I);
}

void orc_tsdf_integrator_set_layer(orc_tsdf_integrator* I, orc_tsdf_layer* layer) {
  I->layer = layer;
}

/* ---- ThreadSafeIndex [recalled, voxblox utils/ thread-safe index]: the order points are visited in ------ */
typedef struct {
  uint32_t key; /* bit pattern of the f32 squared norm (non-negative: orders like the float; NaN last) */
  int64_t idx;
} sorted_entry;

static int sorted_cmp(const void* a, const void* b) {
  const sorted_entry* x = (const sorted_entry*)a;
  const sorted_entry* y = (const sorted_entry*)b;
  if (x->key != y->key) return x->key < y->key ? -1 : 1;
  return x->idx < y->idx ? -1 : (x->idx > y->idx ? 1 : 0); /* the tie rule: ascending point index */
}

/* order[seq] = index of the point visited seq-th; caller frees.  mode 1: MixedThreadSafeIndex (1024-point
 * groups visited round-robin, the tail in order); mode 2: SortedThreadSafeIndex (std::sort by
 * point_C.squaredNorm(), an f32 Eigen reduction (x*x + y*y) + z*z, ascending); else input order. */
static int64_t* visiting_order(int mode, const float* points_C, int64_t n) {
  int64_t* order = (int64_t*)malloc((size_t)(n > 0 ? n : 1) * sizeof(int64_t));
  if (mode == 2) {
    sorted_entry* e = (sorted_entry*)malloc((size_t)(n > 0 ? n : 1) * sizeof(sorted_entry));
    for (int64_t i = 0; i < n; ++i) {
      const float* p = &points_C[3 * i];
      const float sq = (p[0] * p[0] + p[1] * p[1]) + p[2] * p[2];
      memcpy(&e[i].key, &sq, 4);
      e[i].idx = i;
    }
    qsort(e, (size_t)n, sizeof(sorted_entry), sorted_cmp);
    for (int64_t i = 0; i < n; ++i) order[i] = e[i].idx;
    free(e);
    return order;
  }
  const int64_t step_size = 1024;
  const int64_t number_of_groups = n / step_size;
  for (int64_t seq = 0; seq < n; ++seq) {
    int64_t pi = seq;
    if (mode == 1 && seq < number_of_groups * step_size) {
      const int64_t group_num = seq % number_of_groups;
      const int64_t position_in_group = seq / number_of_groups;
      pi = group_num * step_size + position_in_group;
    }
    order[seq] = pi;
  }
  return order;
}

/* ---- the single thread's own event log: the same events, in the same format, the racing kernel logs
 * (include/voxgraph_amd_bench.h) -- every update its own one-record fold.  A sequential run is one legal interleaving, so
 * tsdf_replay.c must accept it; tests/test_tsdf_replay_cpu.py checks that, and that it rejects the log once tampered with. */
void orc_tsdf_integrator_set_log(orc_tsdf_integrator* I, uint64_t* buffer, int64_t capacity_words) {
  I->log = buffer;
  I->log_cap = buffer ? capacity_words : 0;
  I->log_n = 0;
  I->log_lost = 0;
}

int64_t orc_tsdf_integrator_log_words(const orc_tsdf_integrator* I, int64_t* lost) {
  if (lost) *lost = I->log_lost;
  return I->log_n;
}

void orc_tsdf_integrator_download_sets(const orc_tsdf_integrator* I, uint64_t* start_set, uint64_t* observed_set,
                                       uint64_t offsets[2]) {
  if (start_set) memcpy(start_set, I->start_set.slots, ORC_SET_SIZE * sizeof(uint64_t));
  if (observed_set) memcpy(observed_set, I->observed_set.slots, ORC_SET_SIZE * sizeof(uint64_t));
  if (offsets) {
    offsets[0] = I->start_set.offset;
    offsets[1] = I->observed_set.offset;
  }
}

static uint64_t* log_reserve(orc_tsdf_integrator* I, int64_t words) {
  if (!I->log) return NULL;
  if (I->log_n + words > I->log_cap) {
    ++I->log_lost;
    return NULL;
  }
  uint64_t* w = I->log + I->log_n;
  I->log_n += words;
  return w;
}

static void log4(orc_tsdf_integrator* I, uint64_t a, uint64_t b, uint64_t c, uint64_t d) {
  uint64_t* w = log_reserve(I, 4);
  if (w) {
    w[0] = a; w[1] = b; w[2] = c; w[3] = d;
  }
}

static uint64_t log_word(float d, float w) {
  uint32_t a, b;
  memcpy(&a, &d, 4);
  memcpy(&b, &w, 4);
  return (uint64_t)a | ((uint64_t)b << 32);
}

static uint32_t log_rgba(const uint8_t* c) {
  return (uint32_t)c[0] | ((uint32_t)c[1] << 8) | ((uint32_t)c[2] << 16) | ((uint32_t)c[3] << 24);
}

int64_t orc_tsdf_integrate(orc_tsdf_integrator* I, const float T_G_C[7], const float* points_C,
                           const uint8_t* rgba, int64_t n, int freespace_points) {
  const orc_tsdf_config* c = &I->cfg;
  orc_tsdf_layer* L = I->layer;
  const int vps = L->vps;
  const float vsi = L->voxel_size_inv;
  int64_t updates = 0;
  static const uint8_t zero_color[4] = {0, 0, 0, 0};

  if ((++I->reset_counter) >= c->clear_checks_every_n_frames) {
    I->reset_counter = 0;
    approx_set_reset(&I->start_set);
    approx_set_reset(&I->observed_set);
  }
  int64_t* order = visiting_order(c->integration_order, points_C, n);
  const float origin[3] = {T_G_C[4], T_G_C[5], T_G_C[6]};

  for (int64_t seq = 0; seq < n; ++seq) {
    const int64_t pi = order[seq];
    const float* point_C = &points_C[3 * pi];
    const uint8_t* color = rgba ? &rgba[4 * pi] : zero_color;
    int is_clearing = 0;
    if (!point_is_valid(c, point_C, freespace_points, &is_clearing)) continue;
    float point_G[3];
    transform_point(T_G_C, point_C, point_G);
    /* start-voxel dedup on a grid start_voxel_subsampling_factor times finer */
    int64_t gidx[3];
    start_cell(c, vsi, point_G, gidx);
    if (!I->log) {
      if (!approx_set_replace(&I->start_set, gidx)) continue;
    } else {
      uint64_t value, returned;
      const int is_new = approx_set_exchange(&I->start_set, gidx, &value, &returned);
      log4(I, 1, (uint64_t)pi, value, returned);
      if (!is_new) continue;
    }

    orc_ray ray;
    fast_ray_setup(c, vsi, origin, point_G, is_clearing, &ray);
    if (I->log) log4(I, 3 | (ray.bad ? 0x100u : 0u), (uint64_t)pi, ray.bad ? 0 : (uint64_t)(ray.ray_length_in_steps + 1), 0);
    if (ray.bad) continue;
    int64_t consecutive_ray_collisions = 0;
    for (int64_t current_step = 0; current_step <= ray.ray_length_in_steps; ++current_step) {
      int64_t v[3];
      ray_next(&ray, v);

      int is_new;
      if (!I->log) {
        is_new = approx_set_replace(&I->observed_set, v);
      } else {
        uint64_t value, returned;
        is_new = approx_set_exchange(&I->observed_set, v, &value, &returned);
        log4(I, 4 | ((uint64_t)current_step << 8), (uint64_t)pi, value, returned);
      }
      if (!is_new) {
        ++consecutive_ray_collisions;
      } else {
        consecutive_ray_collisions = 0;
      }
      if (consecutive_ray_collisions > c->max_consecutive_ray_collisions) break;
      /* allocateStorageAndGetVoxelPtr: getBlockIndexFromGlobalVoxelIndex + local index */
      int32_t b[3], lv[3];
      for (int a = 0; a < 3; ++a) {
        int64_t q = v[a] / vps, r = v[a] % vps;
        if (r < 0) {
          r += vps;
          q -= 1;
        }
        b[a] = (int32_t)q;
        lv[a] = (int32_t)r;
      }
      int slot = layer_get_or_allocate(L, b[0], b[1], b[2]);
      size_t lin = (size_t)lv[0] + (size_t)vps * ((size_t)lv[1] + (size_t)vps * (size_t)lv[2]);
      size_t at = (size_t)slot * L->nvox + lin;
      const uint64_t word_before = log_word(L->distance[at], L->weight[at]);
      const uint32_t colour_before = log_rgba(&L->rgba[4 * at]);
      update_voxel(c, L->voxel_size, origin, point_G, v, color, point_weight(c, point_C), &L->distance[at], &L->weight[at],
                   &L->rgba[4 * at]);
      ++updates;
      if (I->log) {
        uint64_t* w = log_reserve(I, 7);
        if (w) {
          const uint64_t word_after = log_word(L->distance[at], L->weight[at]);
          const uint32_t colour_after = log_rgba(&L->rgba[4 * at]);
          const uint64_t flags = (word_after != word_before ? 1u : 0u) | (colour_after != colour_before ? 2u : 0u);
          w[0] = 5 | (1ull << 8) | (flags << 40);
          w[1] = pack_voxel_key(v, 0);
          w[2] = word_before;
          w[3] = word_after;
          w[4] = (uint64_t)colour_before | ((uint64_t)colour_after << 32);
          w[5] = (uint64_t)at;
          w[6] = (uint64_t)pi | ((uint64_t)current_step << 32);
        }
      }
    }
  }
  free(order);
  return updates;
}

int64_t orc_tsdf_integrate_sequence(orc_tsdf_integrator* I, int n_scans, const float* poses,
                                    const float* points_C, int64_t n, int repeats) {
  int64_t updates = 0;
  for (int r = 0; r < repeats; ++r)
    for (int k = 0; k < n_scans; ++k)
      updates += orc_tsdf_integrate(I, &poses[7 * k], &points_C[(size_t)3 * n * k], NULL, n, 0);
  return updates;
}

/* ---- MergedTsdfIntegrator ---------------------------------------------------------------------- */
typedef struct {
  uint64_t key;  /* bit 63: clearing ray; bits 62..0: end voxel, 21 bits per axis biased by 2^20 */
  int64_t rank;  /* position in the visiting order (MixedThreadSafeIndex) */
  int64_t pi;    /* point index */
} merged_entry;

static int merged_cmp(const void* a, const void* b) {
  const merged_entry* x = (const merged_entry*)a;
  const merged_entry* y = (const merged_entry*)b;
  if (x->key != y->key) return x->key < y->key ? -1 : 1;
  return x->rank < y->rank ? -1 : (x->rank > y->rank ? 1 : 0);
}

static uint64_t pack_voxel_key(const int64_t v[3], int clearing) {
  const uint64_t B = 1u << 20;
  return ((uint64_t)clearing << 63) | (((uint64_t)(v[0] + (int64_t)B) & 0x1fffff) << 42) |
         (((uint64_t)(v[1] + (int64_t)B) & 0x1fffff) << 21) | ((uint64_t)(v[2] + (int64_t)B) & 0x1fffff);
}

/* is `key` (a surface key) the end voxel of some surface group?  keys[0..n_surface) sorted */
static int in_voxel_map(const merged_entry* e, int64_t n_surface, uint64_t key) {
  int64_t lo = 0, hi = n_surface;
  while (lo < hi) {
    int64_t mid = (lo + hi) / 2;
    if (e[mid].key < key) lo = mid + 1; else hi = mid;
  }
  return lo < n_surface && e[lo].key == key;
}

int64_t orc_tsdf_merged_integrate(orc_tsdf_integrator* I, const float T_G_C[7], const float* points_C,
                                  const uint8_t* rgba, int64_t n, int freespace_points) {
  const orc_tsdf_config* c = &I->cfg;
  orc_tsdf_layer* L = I->layer;
  const int vps = L->vps;
  const float vsi = L->voxel_size_inv;
  static const uint8_t zero_color[4] = {0, 0, 0, 0};
  const float origin[3] = {T_G_C[4], T_G_C[5], T_G_C[6]};
  merged_entry* e = (merged_entry*)malloc((size_t)(n > 0 ? n : 1) * sizeof(merged_entry));
  int64_t m = 0, n_surface = 0, updates = 0;
  /* bundleRays */
  int64_t* order = visiting_order(c->integration_order, points_C, n);
  for (int64_t seq = 0; seq < n; ++seq) {
    const int64_t pi = order[seq];
    const float* point_C = &points_C[3 * pi];
    int is_clearing;
    const float ray_distance = norm3(point_C);
    if (ray_distance < c->min_ray_length_m) {
      continue;
    } else if (ray_distance > c->max_ray_length_m) {
      if (c->allow_clear || freespace_points) is_clearing = 1; else continue;
    } else {
      is_clearing = freespace_points;
    }
    float point_G[3];
    transform_point(T_G_C, point_C, point_G);
    int64_t v[3];
    for (int a = 0; a < 3; ++a) v[a] = grid_index(point_G[a] * vsi + kCoordinateEpsilon);
    e[m].key = pack_voxel_key(v, is_clearing);
    e[m].rank = seq;
    e[m].pi = pi;
    ++m;
    if (!is_clearing) ++n_surface;
  }
  free(order);
  qsort(e, (size_t)m, sizeof(merged_entry), merged_cmp);
  /* integrateRays(clearing_ray = false), then (true): the sort put the surface groups first */
  for (int64_t i0 = 0; i0 < m;) {
    int64_t i1 = i0;
    while (i1 < m && e[i1].key == e[i0].key) ++i1;
    const int clearing_ray = (int)(e[i0].key >> 63);
    /* integrateVoxel: merge */
    float merged_point_C[3] = {0.0f, 0.0f, 0.0f}, merged_weight = 0.0f;
    uint8_t merged_color[4] = {0, 0, 0, 0};
    for (int64_t i = i0; i < i1; ++i) {
      const float* point_C = &points_C[3 * e[i].pi];
      const uint8_t* color = rgba ? &rgba[4 * e[i].pi] : zero_color;
      float point_weight;
      if (c->use_const_weight) {
        point_weight = 1.0f;
      } else {
        float dist_z = fabsf(point_C[2]);
        point_weight = dist_z > kEpsilon ? 1.0f / (dist_z * dist_z) : 0.0f;
      }
      if (point_weight < kEpsilon) continue;
      const float total = merged_weight + point_weight;
      for (int a = 0; a < 3; ++a)
        merged_point_C[a] = (merged_point_C[a] * merged_weight + point_C[a] * point_weight) / total;
      /* Color::blendTwoColors(merged_color, merged_weight, color, point_weight) */
      {
        const float fw = merged_weight / total, sw = point_weight / total;
        for (int k = 0; k < 4; ++k)
          merged_color[k] = (uint8_t)roundf((float)merged_color[k] * fw + (float)color[k] * sw);
      }
      merged_weight += point_weight;
      if (clearing_ray) break; /* only take first point when clearing */
    }
    const uint64_t own_key = e[i0].key;
    i0 = i1;
    if (merged_weight == 0.0f) continue; /* every update would leave its voxel unchanged */
    float point_G[3];
    transform_point(T_G_C, merged_point_C, point_G);
    /* RayCaster(origin, merged_point_G, clearing_ray, carving, max_ray, voxel_size_inv, trunc) */
    float d[3] = {point_G[0] - origin[0], point_G[1] - origin[1], point_G[2] - origin[2]};
    float len = norm3(d);
    float unit_ray[3] = {d[0] / len, d[1] / len, d[2] / len};
    float ray_start[3], ray_end[3];
    const float trunc = c->default_truncation_distance;
    if (clearing_ray) {
      float ray_length = fminf(fmaxf(len - trunc, 0.0f), c->max_ray_length_m);
      for (int a = 0; a < 3; ++a) {
        ray_end[a] = origin[a] + unit_ray[a] * ray_length;
        ray_start[a] = c->voxel_carving_enabled ? origin[a] : ray_end[a];
      }
    } else {
      for (int a = 0; a < 3; ++a) {
        ray_end[a] = point_G[a] + unit_ray[a] * trunc;
        ray_start[a] = c->voxel_carving_enabled ? origin[a] : (point_G[a] - unit_ray[a] * trunc);
      }
    }
    float start_scaled[3], end_scaled[3];
    int bad = 0;
    /* cast_from_origin == true (the RayCaster default, which MergedTsdfIntegrator::integrateVoxel
     * leaves alone; only the fast integrator passes false): setupRayCaster(start_scaled, end_scaled),
     * the walk runs from the sensor outwards */
    for (int a = 0; a < 3; ++a) {
      start_scaled[a] = ray_start[a] * vsi;
      end_scaled[a] = ray_end[a] * vsi;
      if (isnan(start_scaled[a]) || isnan(end_scaled[a])) bad = 1;
    }
    if (bad) continue;
    int64_t curr[3], ray_length_in_steps = 0;
    int step_sign[3];
    float t_to_next[3], t_step[3];
    for (int a = 0; a < 3; ++a) {
      curr[a] = grid_index(start_scaled[a] + kCoordinateEpsilon);
      int64_t end_index = grid_index(end_scaled[a] + kCoordinateEpsilon);
      int64_t diff = end_index - curr[a];
      ray_length_in_steps += diff < 0 ? -diff : diff;
      float ray_scaled = end_scaled[a] - start_scaled[a];
      step_sign[a] = signum(ray_scaled);
      float corrected_step = (float)(step_sign[a] > 0 ? step_sign[a] : 0);
      float distance_to_boundary = corrected_step - (start_scaled[a] - (float)curr[a]);
      if (ray_scaled == 0.0f) {
        t_to_next[a] = INFINITY;
        t_step[a] = INFINITY;
      } else {
        t_to_next[a] = distance_to_boundary / ray_scaled;
        t_step[a] = (float)step_sign[a] / ray_scaled;
      }
    }
    for (int64_t current_step = 0; current_step <= ray_length_in_steps; ++current_step) {
      int64_t v[3] = {curr[0], curr[1], curr[2]};
      int t_min_idx = 0;
      if (t_to_next[1] < t_to_next[t_min_idx]) t_min_idx = 1;
      if (t_to_next[2] < t_to_next[t_min_idx]) t_min_idx = 2;
      curr[t_min_idx] += step_sign[t_min_idx];
      t_to_next[t_min_idx] += t_step[t_min_idx];
      if (c->enable_anti_grazing) {
        /* skip voxels that are the end voxel of a (different) surface group */
        const uint64_t k = pack_voxel_key(v, 0);
        if ((clearing_ray || k != own_key) && in_voxel_map(e, n_surface, k)) continue;
      }
      int32_t b[3], lv[3];
      for (int a = 0; a < 3; ++a) {
        int64_t q = v[a] / vps, r = v[a] % vps;
        if (r < 0) {
          r += vps;
          q -= 1;
        }
        b[a] = (int32_t)q;
        lv[a] = (int32_t)r;
      }
      int slot = layer_get_or_allocate(L, b[0], b[1], b[2]);
      size_t lin = (size_t)lv[0] + (size_t)vps * ((size_t)lv[1] + (size_t)vps * (size_t)lv[2]);
      size_t at = (size_t)slot * L->nvox + lin;
      update_voxel(c, L->voxel_size, origin, point_G, v, merged_color, merged_weight, &L->distance[at], &L->weight[at],
                   &L->rgba[4 * at]);
      ++updates;
    }
  }
  free(e);
  return updates;
}
