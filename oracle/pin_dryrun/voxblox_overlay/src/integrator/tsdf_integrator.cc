// REHEARSAL STAND-IN (oracle/pin_dryrun/README.md) for voxblox/src/integrator/tsdf_integrator.cc: the classes of the
// overlay header on top of oracle/tsdf_oracle.c -- every scan is integrated by the oracle into its own layer, which is
// then copied into the caller's voxblox::Layer (stand-in).  TEST INFRASTRUCTURE; proves nothing about voxblox.
#include <voxblox/integrator/tsdf_integrator.h>

#include <cstring>

#include "tsdf_oracle.h"

namespace voxblox {
TsdfIntegratorBase::TsdfIntegratorBase(const Config& c, Layer<TsdfVoxel>* layer, bool merged)
    : config_(c), layer_(layer), merged_(merged) {
  orc_tsdf_config o;
  orc_tsdf_config_default(&o);
  o.default_truncation_distance = c.default_truncation_distance;
  o.max_weight = c.max_weight;
  o.voxel_carving_enabled = c.voxel_carving_enabled;
  o.min_ray_length_m = c.min_ray_length_m;
  o.max_ray_length_m = c.max_ray_length_m;
  o.use_const_weight = c.use_const_weight;
  o.allow_clear = c.allow_clear;
  o.use_weight_dropoff = c.use_weight_dropoff;
  o.use_sparsity_compensation_factor = c.use_sparsity_compensation_factor;
  o.sparsity_compensation_factor = c.sparsity_compensation_factor;
  o.start_voxel_subsampling_factor = c.start_voxel_subsampling_factor;
  o.max_consecutive_ray_collisions = c.max_consecutive_ray_collisions;
  o.clear_checks_every_n_frames = c.clear_checks_every_n_frames;
  o.enable_anti_grazing = c.enable_anti_grazing;
  o.integration_order = c.integration_order_mode == "sorted" ? 2 : 1;
  orc_layer_ = orc_tsdf_layer_create(layer->voxel_size(), static_cast<int>(layer->voxels_per_side()));
  orc_ = orc_tsdf_integrator_create(&o, orc_layer_);
}

TsdfIntegratorBase::~TsdfIntegratorBase() {
  orc_tsdf_integrator_destroy(orc_);
  orc_tsdf_layer_destroy(orc_layer_);
}

void TsdfIntegratorBase::integratePointCloud(const Transformation& T_G_C, const Pointcloud& points_C, const Colors& colors,
                                             const bool freespace_points) {
  const auto& q = T_G_C.getRotation();
  const auto& t = T_G_C.getPosition();
  const float T[7] = {q.w(), q.x(), q.y(), q.z(), t[0], t[1], t[2]};
  const size_t n = points_C.size();
  std::vector<float> p(3 * n);
  std::vector<uint8_t> col(4 * n);
  for (size_t i = 0; i < n; ++i) {
    for (int a = 0; a < 3; ++a) p[3 * i + a] = points_C[i][a];
    col[4 * i] = colors[i].r; col[4 * i + 1] = colors[i].g; col[4 * i + 2] = colors[i].b; col[4 * i + 3] = colors[i].a;
  }
  if (merged_) orc_tsdf_merged_integrate(orc_, T, p.data(), col.data(), static_cast<int64_t>(n), freespace_points);
  else orc_tsdf_integrate(orc_, T, p.data(), col.data(), static_cast<int64_t>(n), freespace_points);
  // the oracle's layer -> the caller's
  const int nb = orc_tsdf_layer_num_blocks(orc_layer_);
  const size_t vox = layer_->voxels_per_side() * layer_->voxels_per_side() * layer_->voxels_per_side();
  std::vector<int32_t> bi(3 * static_cast<size_t>(nb));
  std::vector<float> d(nb * vox), w(nb * vox);
  std::vector<uint8_t> rgba(4 * nb * vox);
  orc_tsdf_layer_download(orc_layer_, bi.data(), d.data(), w.data(), rgba.data());
  for (int b = 0; b < nb; ++b) {
    BlockIndex index;
    for (int a = 0; a < 3; ++a) index[a] = bi[3 * b + a];
    auto block = layer_->allocateBlockPtrByIndex(index);
    for (size_t l = 0; l < vox; ++l) {
      TsdfVoxel& v = block->getVoxelByLinearIndex(l);
      v.distance = d[b * vox + l];
      v.weight = w[b * vox + l];
      const uint8_t* c = &rgba[4 * (b * vox + l)];
      v.color = Color(c[0], c[1], c[2], c[3]);
    }
  }
}
}  // namespace voxblox
