/* oracle/iso_oracle.c -- see iso_oracle.h.  TEST INFRASTRUCTURE, parity unpinned.
 * Build with -ffp-contract=off. */
#include "iso_oracle.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>

#include "reg_oracle.h"

typedef struct {
  int64_t key[3];
  int used;
} cell_slot;

static uint64_t cell_hash(const int64_t k[3]) {
  uint64_t h = (uint64_t)k[0] * 0x9E3779B97F4A7C15ull;
  h ^= (uint64_t)k[1] * 0xC2B2AE3D27D4EB4Full + (h << 6) + (h >> 2);
  h ^= (uint64_t)k[2] * 0x165667B19E3779F9ull + (h << 6) + (h >> 2);
  return h;
}

int64_t orc_isosurface_points(float voxel_size, int vps, int n_blocks,
                              const int32_t* block_index, const float* tsdf_distance,
                              const float* tsdf_weight, float min_weight, float* xyz,
                              float* distance, float* weight) {
  const size_t nvox = (size_t)vps * vps * vps;
  if (n_blocks == 0) return 0;
  /* layers for the interpolator: validity = weight > 0 (Interpolator<TsdfVoxel>) */
  uint8_t* valid = (uint8_t*)malloc((size_t)n_blocks * nvox);
  if (!valid) return -1;
  for (size_t i = 0; i < (size_t)n_blocks * nvox; ++i) valid[i] = tsdf_weight[i] > 0.0f;
  orc_layer Ld, Lw;
  if (orc_layer_init(&Ld, voxel_size, vps, n_blocks, block_index, tsdf_distance, valid) != 0 ||
      orc_layer_init(&Lw, voxel_size, vps, n_blocks, block_index, tsdf_weight, valid) != 0)
    return -1;
  const float block_size = (float)vps * voxel_size;

  /* voxel lookup through the layer's block table: returns flat index or -1 */
#define VOX_AT(bx, by, bz, vx, vy, vz, out)                                                  \
  do {                                                                                       \
    int32_t bb[3] = {bx, by, bz};                                                            \
    int vv[3] = {vx, vy, vz};                                                                \
    for (int a_ = 0; a_ < 3; ++a_) {                                                         \
      if (vv[a_] < 0) { vv[a_] += vps; bb[a_]--; }                                           \
      if (vv[a_] >= vps) { vv[a_] -= vps; bb[a_]++; }                                        \
    }                                                                                        \
    int32_t r_[3] = {bb[0] - Ld.lut_min[0], bb[1] - Ld.lut_min[1], bb[2] - Ld.lut_min[2]};   \
    (out) = -1;                                                                              \
    if (r_[0] >= 0 && r_[1] >= 0 && r_[2] >= 0 && r_[0] < Ld.lut_dim[0] &&                   \
        r_[1] < Ld.lut_dim[1] && r_[2] < Ld.lut_dim[2]) {                                    \
      int32_t s_ = Ld.lut[(size_t)r_[0] + (size_t)Ld.lut_dim[0] *                            \
                                              ((size_t)r_[1] + (size_t)Ld.lut_dim[1] * (size_t)r_[2])]; \
      if (s_ >= 0)                                                                           \
        (out) = (int64_t)s_ * (int64_t)nvox + vv[0] + vps * (vv[1] + vps * vv[2]);           \
    }                                                                                        \
  } while (0)

  /* dedup table: 0.5-voxel cells already taken (getConnectedMesh) */
  size_t cap = 1024;
  cell_slot* table = (cell_slot*)calloc(cap, sizeof(cell_slot));
  size_t used = 0;
  const float threshold = (float)(0.5 * (double)voxel_size);
  const double threshold_inv = 1.0 / (double)threshold;
  int64_t n_out = 0;

  for (int b = 0; b < n_blocks; ++b) {
    const int32_t* bi = &block_index[3 * b];
    for (size_t lin = 0; lin < nvox; ++lin) {
      const int v[3] = {(int)(lin % (size_t)vps), (int)((lin / (size_t)vps) % (size_t)vps),
                        (int)(lin / ((size_t)vps * vps))};
      const size_t at0 = (size_t)b * nvox + lin;
      const float w0 = tsdf_weight[at0], s0 = tsdf_distance[at0];
      if (!(w0 > min_weight)) continue;
      for (int axis = 0; axis < 3; ++axis) {
        int e[3] = {0, 0, 0};
        e[axis] = 1;
        int64_t at1;
        VOX_AT(bi[0], bi[1], bi[2], v[0] + e[0], v[1] + e[1], v[2] + e[2], at1);
        if (at1 < 0 || !(tsdf_weight[at1] > min_weight)) continue;
        const float s1 = tsdf_distance[at1];
        /* MarchingCubes::interpolateEdgeVertices: only edges with a zero crossing */
        if ((s0 < 0.0f) == (s1 < 0.0f)) continue;
        /* the edge must belong to at least one fully observed dual cell */
        const int ob = (axis + 1) % 3, oc = (axis + 2) % 3;
        int any_cell = 0;
        for (int sb = 0; sb < 2 && !any_cell; ++sb)
          for (int sc = 0; sc < 2 && !any_cell; ++sc) {
            int base[3] = {v[0], v[1], v[2]};
            base[ob] -= sb;
            base[oc] -= sc;
            int all = 1;
            for (int k = 0; k < 8 && all; ++k) {
              int64_t atc;
              VOX_AT(bi[0], bi[1], bi[2], base[0] + (k & 1), base[1] + ((k >> 1) & 1),
                     base[2] + ((k >> 2) & 1), atc);
              if (atc < 0 || !(tsdf_weight[atc] > min_weight)) all = 0;
            }
            any_cell = all;
          }
        if (!any_cell) continue;
        /* MarchingCubes::interpolateVertex, low -> high along the edge */
        float p0[3], vert[3];
        for (int a = 0; a < 3; ++a)
          p0[a] = (float)bi[a] * block_size + (float)(((double)(float)v[a] + 0.5) * (double)voxel_size);
        float p1a;
        {
          /* centre of the neighbour voxel along `axis` (may live in the next block) */
          int nv = v[axis] + 1;
          int32_t nb = bi[axis];
          if (nv >= vps) { nv -= vps; nb++; }
          p1a = (float)nb * block_size + (float)(((double)(float)nv + 0.5) * (double)voxel_size);
        }
        const float sdf_diff = s0 - s1;
        for (int a = 0; a < 3; ++a) vert[a] = p0[a];
        if (fabsf(sdf_diff) >= 1e-6f) {
          const float t = s0 / sdf_diff;
          vert[axis] = p0[axis] + t * (p1a - p0[axis]);
        } else {
          vert[axis] = 0.5f * (p0[axis] + p1a);
        }
        /* getConnectedMesh: first vertex per round(v / threshold) cell wins */
        int64_t key[3];
        for (int a = 0; a < 3; ++a) key[a] = (int64_t)round((double)vert[a] * threshold_inv);
        if ((used + 1) * 2 > cap) {
          size_t ncap = cap * 2;
          cell_slot* nt = (cell_slot*)calloc(ncap, sizeof(cell_slot));
          if (!nt) return -1;
          for (size_t i = 0; i < cap; ++i)
            if (table[i].used) {
              size_t h = cell_hash(table[i].key) & (ncap - 1);
              while (nt[h].used) h = (h + 1) & (ncap - 1);
              nt[h] = table[i];
            }
          free(table);
          table = nt;
          cap = ncap;
        }
        size_t h = cell_hash(key) & (cap - 1);
        int dup = 0;
        while (table[h].used) {
          if (table[h].key[0] == key[0] && table[h].key[1] == key[1] && table[h].key[2] == key[2]) {
            dup = 1;
            break;
          }
          h = (h + 1) & (cap - 1);
        }
        if (dup) continue;
        table[h].used = 1;
        memcpy(table[h].key, key, sizeof(key));
        ++used;
        /* Interpolator::getVoxel(vertex, &voxel, true): trilinear distance and weight */
        float d8[8], w8[8], q[8], q2[8];
        if (!orc_get_voxels_and_q(&Ld, vert, d8, q)) continue;
        if (!orc_get_voxels_and_q(&Lw, vert, w8, q2)) continue;
        static const float B[8][8] = {
            {1, 0, 0, 0, 0, 0, 0, 0},   {-1, 0, 0, 0, 1, 0, 0, 0},  {-1, 0, 1, 0, 0, 0, 0, 0},
            {-1, 1, 0, 0, 0, 0, 0, 0},  {1, 0, -1, 0, -1, 0, 1, 0}, {1, -1, -1, 1, 0, 0, 0, 0},
            {1, -1, 0, 0, -1, 1, 0, 0}, {-1, 1, 1, -1, 1, -1, -1, 1}};
        float di = 0.0f, wi = 0.0f;
        for (int r = 0; r < 8; ++r) {
          float cd = 0.0f, cw = 0.0f;
          for (int k = 0; k < 8; ++k) {
            cd += B[r][k] * d8[k];
            cw += B[r][k] * w8[k];
          }
          di += q[r] * cd;
          wi += q[r] * cw;
        }
        if (xyz) {
          xyz[3 * n_out] = vert[0];
          xyz[3 * n_out + 1] = vert[1];
          xyz[3 * n_out + 2] = vert[2];
          distance[n_out] = di;
          weight[n_out] = wi;
        }
        ++n_out;
      }
    }
  }
  free(table);
  free(valid);
  orc_layer_free(&Ld);
  orc_layer_free(&Lw);
  return n_out;
}
