// voxblox::EsdfMap: a Layer<EsdfVoxel> with its config ([recalled]).  TEST INFRASTRUCTURE.
#ifndef ORACLE_REF_SHIMS_VOXBLOX_CORE_ESDF_MAP_H_
#define ORACLE_REF_SHIMS_VOXBLOX_CORE_ESDF_MAP_H_
#include "voxblox/core/layer.h"
namespace voxblox {
class EsdfMap {
 public:
  typedef std::shared_ptr<EsdfMap> Ptr;
  struct Config {
    FloatingPoint esdf_voxel_size = 0.2;
    size_t esdf_voxels_per_side = 16u;
  };
  explicit EsdfMap(const Config& c) : layer_(new Layer<EsdfVoxel>(c.esdf_voxel_size, c.esdf_voxels_per_side)) {}
  const Layer<EsdfVoxel>& getEsdfLayer() const { return *layer_; }
  Layer<EsdfVoxel>* getEsdfLayerPtr() { return layer_.get(); }
  FloatingPoint block_size() const { return layer_->block_size(); }
  FloatingPoint voxel_size() const { return layer_->voxel_size(); }

 private:
  std::unique_ptr<Layer<EsdfVoxel>> layer_;
};
}  // namespace voxblox
#endif
