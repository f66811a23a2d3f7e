/* oracle/esdf_oracle.c -- see esdf_oracle.h.  TEST INFRASTRUCTURE, parity unpinned,
 * [recalled] from voxblox.  Build with -ffp-contract=off. */
#include "esdf_oracle.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>

void orc_esdf_config_default(orc_esdf_config* c) {
  c->max_distance_m = 2.0f;
  c->min_distance_m = 0.2f;
  c->default_distance_m = 2.0f;
  c->min_diff_m = 0.001f;
  c->min_weight = 1e-6f;
  c->num_buckets = 20;
}

/* ---- BucketQueue<GlobalIndex> [recalled]: FIFO queues indexed by |value| -------- */
typedef struct {
  int64_t* data;
  size_t head, tail, cap;
} fifo;

static int fifo_push(fifo* q, int64_t v) {
  if (q->tail == q->cap) {
    if (q->head > q->cap / 2) {
      memmove(q->data, q->data + q->head, (q->tail - q->head) * sizeof(int64_t));
      q->tail -= q->head;
      q->head = 0;
    } else {
      size_t cap = q->cap ? 2 * q->cap : 1024;
      int64_t* d = (int64_t*)realloc(q->data, cap * sizeof(int64_t));
      if (!d) return -1;
      q->data = d;
      q->cap = cap;
    }
  }
  q->data[q->tail++] = v;
  return 0;
}

typedef struct {
  fifo* buckets;
  int num_buckets, last_bucket_index;
  double max_val;
  size_t num_elements;
} bucket_queue;

static int bq_push(bucket_queue* q, int64_t key, double value) {
  if (value > q->max_val) value = q->max_val;
  int bucket_index = (int)floor(fabs(value) / q->max_val * (q->num_buckets - 1));
  if (bucket_index >= q->num_buckets) bucket_index = q->num_buckets - 1;
  if (bucket_index < q->last_bucket_index) q->last_bucket_index = bucket_index;
  q->num_elements++;
  return fifo_push(&q->buckets[bucket_index], key);
}

static int64_t bq_pop_front(bucket_queue* q) {
  while (q->last_bucket_index < q->num_buckets &&
         q->buckets[q->last_bucket_index].head == q->buckets[q->last_bucket_index].tail)
    q->last_bucket_index++;
  fifo* f = &q->buckets[q->last_bucket_index];
  q->num_elements--;
  return f->data[f->head++];
}

int64_t orc_esdf_from_tsdf_batch(const orc_esdf_config* cfg, float voxel_size, int vps,
                                 int n_blocks, const int32_t* block_index,
                                 const float* tsdf_distance, const float* tsdf_weight,
                                 float* esdf_distance, uint8_t* esdf_observed) {
  const size_t nvox = (size_t)vps * vps * vps, total = (size_t)n_blocks * nvox;
  memset(esdf_distance, 0, total * sizeof(float));
  memset(esdf_observed, 0, total);
  if (n_blocks == 0) return 0;
  /* dense block table (stands in for the layer's hash map) */
  int32_t mn[3], mx[3];
  for (int a = 0; a < 3; ++a) mn[a] = mx[a] = block_index[a];
  for (int b = 1; b < n_blocks; ++b)
    for (int a = 0; a < 3; ++a) {
      int32_t v = block_index[3 * b + a];
      if (v < mn[a]) mn[a] = v;
      if (v > mx[a]) mx[a] = v;
    }
  int32_t dim[3] = {mx[0] - mn[0] + 1, mx[1] - mn[1] + 1, mx[2] - mn[2] + 1};
  size_t cells = (size_t)dim[0] * dim[1] * dim[2];
  int32_t* lut = (int32_t*)malloc(cells * sizeof(int32_t));
  uint8_t* fixed = (uint8_t*)calloc(total, 1);
  uint8_t* in_queue = (uint8_t*)calloc(total, 1);
  bucket_queue q;
  q.num_buckets = cfg->num_buckets;
  q.last_bucket_index = 0;
  q.max_val = cfg->max_distance_m;
  q.num_elements = 0;
  q.buckets = (fifo*)calloc((size_t)cfg->num_buckets, sizeof(fifo));
  if (!lut || !fixed || !in_queue || !q.buckets) return -1;
  for (size_t i = 0; i < cells; ++i) lut[i] = -1;
  for (int b = 0; b < n_blocks; ++b)
    lut[(size_t)(block_index[3 * b] - mn[0]) +
        (size_t)dim[0] * ((size_t)(block_index[3 * b + 1] - mn[1]) +
                          (size_t)dim[1] * (size_t)(block_index[3 * b + 2] - mn[2]))] = b;

  /* updateFromTsdfBlocks(..., incremental = false) */
  for (int b = 0; b < n_blocks; ++b)
    for (size_t lin = 0; lin < nvox; ++lin) {
      size_t at = (size_t)b * nvox + lin;
      if (tsdf_weight[at] < cfg->min_weight) continue; /* stays unobserved */
      float d = tsdf_distance[at];
      esdf_observed[at] = 1;
      if (fabsf(d) < cfg->min_distance_m) { /* isFixed */
        esdf_distance[at] = d;
        fixed[at] = 1;
        in_queue[at] = 1;
        if (bq_push(&q, (int64_t)at, d) != 0) return -1;
      } else {
        float sgn = (float)((d > 0.0f) - (d < 0.0f));
        esdf_distance[at] = sgn * cfg->default_distance_m;
      }
    }

  /* Neighborhood<Connectivity::kTwentySix>: 6 faces, 12 edges, 8 corners */
  int off[26][3];
  float step[26];
  int k = 0;
  for (int pass = 1; pass <= 3; ++pass)
    for (int dx = -1; dx <= 1; ++dx)
      for (int dy = -1; dy <= 1; ++dy)
        for (int dz = -1; dz <= 1; ++dz) {
          int nz = (dx != 0) + (dy != 0) + (dz != 0);
          if (nz != pass) continue;
          off[k][0] = dx; off[k][1] = dy; off[k][2] = dz;
          step[k] = sqrtf((float)nz);
          ++k;
        }

  /* processOpenSet */
  int64_t num_updates = 0;
  while (q.num_elements > 0) {
    size_t at = (size_t)bq_pop_front(&q);
    in_queue[at] = 0;
    float vd = esdf_distance[at];
    if (!esdf_observed[at] || vd >= cfg->max_distance_m || vd <= -cfg->max_distance_m) continue;
    int b = (int)(at / nvox);
    size_t lin = at % nvox;
    int v[3] = {(int)(lin % (size_t)vps), (int)((lin / (size_t)vps) % (size_t)vps),
                (int)(lin / ((size_t)vps * vps))};
    for (int n = 0; n < 26; ++n) {
      int32_t nb[3];
      int nv[3];
      for (int a = 0; a < 3; ++a) {
        nb[a] = block_index[3 * b + a];
        nv[a] = v[a] + off[n][a];
        if (nv[a] < 0) { nv[a] += vps; nb[a]--; }
        if (nv[a] >= vps) { nv[a] -= vps; nb[a]++; }
      }
      int32_t r[3] = {nb[0] - mn[0], nb[1] - mn[1], nb[2] - mn[2]};
      if (r[0] < 0 || r[1] < 0 || r[2] < 0 || r[0] >= dim[0] || r[1] >= dim[1] || r[2] >= dim[2]) continue;
      int32_t slot = lut[(size_t)r[0] + (size_t)dim[0] * ((size_t)r[1] + (size_t)dim[1] * (size_t)r[2])];
      if (slot < 0) continue;
      size_t nat = (size_t)slot * nvox + (size_t)nv[0] + (size_t)vps * ((size_t)nv[1] + (size_t)vps * (size_t)nv[2]);
      if (!esdf_observed[nat] || fixed[nat]) continue;
      float nd = esdf_distance[nat];
      float dist_to_neighbor = step[n] * voxel_size;
      if (vd > 0.0f && nd > 0.0f) {
        if (vd + dist_to_neighbor + cfg->min_diff_m < nd) {
          esdf_distance[nat] = vd + dist_to_neighbor;
          ++num_updates;
          if (!in_queue[nat]) {
            in_queue[nat] = 1;
            if (bq_push(&q, (int64_t)nat, esdf_distance[nat]) != 0) return -1;
          }
        }
      } else if (vd < 0.0f && nd < 0.0f) {
        if (vd - dist_to_neighbor - cfg->min_diff_m > nd) {
          esdf_distance[nat] = vd - dist_to_neighbor;
          ++num_updates;
          if (!in_queue[nat]) {
            in_queue[nat] = 1;
            if (bq_push(&q, (int64_t)nat, esdf_distance[nat]) != 0) return -1;
          }
        }
      }
    }
  }
  for (int i = 0; i < cfg->num_buckets; ++i) free(q.buckets[i].data);
  free(q.buckets);
  free(lut);
  free(fixed);
  free(in_queue);
  return num_updates;
}
