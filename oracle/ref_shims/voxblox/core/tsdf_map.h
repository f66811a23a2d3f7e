// voxblox::TsdfMap: a Layer<TsdfVoxel> with its config ([recalled]).  TEST INFRASTRUCTURE.
#ifndef ORACLE_REF_SHIMS_VOXBLOX_CORE_TSDF_MAP_H_
#define ORACLE_REF_SHIMS_VOXBLOX_CORE_TSDF_MAP_H_
#include "voxblox/core/layer.h"
namespace voxblox {
class TsdfMap {
 public:
  typedef std::shared_ptr<TsdfMap> Ptr;
  struct Config {
    FloatingPoint tsdf_voxel_size = 0.2;
    size_t tsdf_voxels_per_side = 16u;
  };
  explicit TsdfMap(const Config& c) : layer_(new Layer<TsdfVoxel>(c.tsdf_voxel_size, c.tsdf_voxels_per_side)) {}
  explicit TsdfMap(const Layer<TsdfVoxel>& layer) : layer_(new Layer<TsdfVoxel>(layer)) {}
  const Layer<TsdfVoxel>& getTsdfLayer() const { return *layer_; }
  Layer<TsdfVoxel>* getTsdfLayerPtr() { return layer_.get(); }
  const Layer<TsdfVoxel>* getTsdfLayerPtr() const { return layer_.get(); }
  FloatingPoint block_size() const { return layer_->block_size(); }
  FloatingPoint voxel_size() const { return layer_->voxel_size(); }

 private:
  std::unique_ptr<Layer<TsdfVoxel>> layer_;
};
}  // namespace voxblox
#endif
