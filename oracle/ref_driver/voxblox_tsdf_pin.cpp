// oracle/PIN.md: voxblox's OWN Fast / Merged TSDF integrators on the sessions of tests/golden/make_tsdf_golden.py
// (TEST INFRASTRUCTURE; built by `make -C oracle pin VOXBLOX=...`).  It needs the real voxblox sources, which this image
// does not have: it is written against voxblox's public API as recalled -- TsdfIntegratorBase::Config,
// FastTsdfIntegrator / MergedTsdfIntegrator(config, Layer<TsdfVoxel>*), integratePointCloud(T_G_C, points_C, colors,
// freespace_points), Layer::getAllAllocatedBlocks / getBlockByIndex, Block::getVoxelByLinearIndex.  Since round 6 it is at
// least COMPILED AND RUN by `make -C oracle pin-dryrun` against a fake checkout whose tsdf_integrator.{h,cc} wrap the oracle
// (oracle/pin_dryrun/README.md); against the real headers the first thing to do is still to make it build.
//
//   voxblox_tsdf_pin sessions.bin out_dir
// reads the sessions (format: make_tsdf_golden.py dump_sessions), integrates each with integrator_threads = 1 in the
// session's integration_order_mode, and writes out_dir/<session>.layer.bin (format: make_tsdf_golden.py compare_layers).
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <string>
#include <vector>

#include <voxblox/core/common.h>
#include <voxblox/core/layer.h>
#include <voxblox/core/voxel.h>
#include <voxblox/integrator/tsdf_integrator.h>

namespace {
template <class T>
bool get(FILE* f, T* v, size_t n = 1) { return fread(v, sizeof(T), n, f) == n; }
}  // namespace

int main(int argc, char** argv) {
  if (argc != 3) {
    fprintf(stderr, "usage: voxblox_tsdf_pin sessions.bin out_dir\n");
    return 2;
  }
  FILE* f = fopen(argv[1], "rb");
  if (!f) return 2;
  int32_t n_sessions = 0;
  if (!get(f, &n_sessions)) return 2;
  for (int s = 0; s < n_sessions; ++s) {
    char name[65] = {0};
    float c[15];
    int32_t merged = 0, n_scans = 0;
    if (!get(f, name, 64) || !get(f, c, 15) || !get(f, &merged) || !get(f, &n_scans)) return 2;
    voxblox::TsdfIntegratorBase::Config cfg;
    cfg.default_truncation_distance = c[0];
    cfg.max_weight = c[1];
    cfg.voxel_carving_enabled = c[2] != 0;
    cfg.min_ray_length_m = c[3];
    cfg.max_ray_length_m = c[4];
    cfg.use_const_weight = c[5] != 0;
    cfg.allow_clear = c[6] != 0;
    cfg.use_weight_dropoff = c[7] != 0;
    cfg.use_sparsity_compensation_factor = c[8] != 0;
    cfg.sparsity_compensation_factor = c[9];
    cfg.start_voxel_subsampling_factor = c[10];
    cfg.max_consecutive_ray_collisions = (int)c[11];
    cfg.clear_checks_every_n_frames = (int)c[12];
    cfg.enable_anti_grazing = c[13] != 0;
    cfg.integration_order_mode = c[14] != 0 ? "sorted" : "mixed";
    cfg.integrator_threads = 1;  // the order oracle/tsdf_oracle.c restates
    voxblox::Layer<voxblox::TsdfVoxel> layer(0.2f, 16);
    voxblox::TsdfIntegratorBase::Ptr integ;
    if (merged) integ.reset(new voxblox::MergedTsdfIntegrator(cfg, &layer));
    else integ.reset(new voxblox::FastTsdfIntegrator(cfg, &layer));
    for (int k = 0; k < n_scans; ++k) {
      float T[7];
      int32_t n = 0;
      if (!get(f, T, 7) || !get(f, &n)) return 2;
      std::vector<float> p(3 * (size_t)n);
      std::vector<uint8_t> col(4 * (size_t)n);
      if (!get(f, p.data(), p.size()) || !get(f, col.data(), col.size())) return 2;
      voxblox::Transformation T_G_C(voxblox::Transformation::Rotation(T[0], T[1], T[2], T[3]),
                                    voxblox::Transformation::Position(T[4], T[5], T[6]));
      voxblox::Pointcloud points_C((size_t)n);
      voxblox::Colors colors((size_t)n);
      for (int i = 0; i < n; ++i) {
        points_C[i] = voxblox::Point(p[3 * i], p[3 * i + 1], p[3 * i + 2]);
        colors[i] = voxblox::Color(col[4 * i], col[4 * i + 1], col[4 * i + 2], col[4 * i + 3]);
      }
      integ->integratePointCloud(T_G_C, points_C, colors, false);
    }
    voxblox::BlockIndexList blocks;
    layer.getAllAllocatedBlocks(&blocks);
    const std::string path = std::string(argv[2]) + "/" + name + ".layer.bin";
    FILE* o = fopen(path.c_str(), "wb");
    if (!o) return 2;
    const int32_t nb = (int32_t)blocks.size();
    fwrite(&nb, 4, 1, o);
    for (const voxblox::BlockIndex& bi : blocks) {
      const voxblox::Block<voxblox::TsdfVoxel>& b = layer.getBlockByIndex(bi);
      const int32_t idx[3] = {bi.x(), bi.y(), bi.z()};
      fwrite(idx, 4, 3, o);
      std::vector<float> d(4096), w(4096);
      std::vector<uint8_t> rgba(4 * 4096);
      for (size_t l = 0; l < 4096; ++l) {
        const voxblox::TsdfVoxel& v = b.getVoxelByLinearIndex(l);
        d[l] = v.distance;
        w[l] = v.weight;
        rgba[4 * l] = v.color.r; rgba[4 * l + 1] = v.color.g; rgba[4 * l + 2] = v.color.b; rgba[4 * l + 3] = v.color.a;
      }
      fwrite(d.data(), 4, 4096, o);
      fwrite(w.data(), 4, 4096, o);
      fwrite(rgba.data(), 1, 4 * 4096, o);
    }
    fclose(o);
    fprintf(stderr, "%s: %d blocks\n", name, nb);
  }
  fclose(f);
  return 0;
}
