// Stand-in for voxgraph::VoxgraphSubmap: the real class derives from cblox::TsdfEsdfSubmap and
// pulls in cblox, voxblox meshing and ROS.  The cost function only needs two layers, a pose, an
// id and the two registration-point samplers -- the sampler and the point struct are the
// reference's own headers.  TEST INFRASTRUCTURE -- see oracle/ref_shims/README.md.
#ifndef ORACLE_REF_SHIMS_VOXGRAPH_SUBMAP_H_
#define ORACLE_REF_SHIMS_VOXGRAPH_SUBMAP_H_
#include <memory>

#include "voxblox/core/layer.h"
#include "voxblox/interpolator/interpolator.h"
// the reference's own files (found through -I /root/reference/voxgraph/include):
#include "voxgraph/frontend/submap_collection/registration_point.h"
#include "voxgraph/frontend/submap_collection/weighted_sampler.h"

namespace voxgraph {
class VoxgraphSubmap {
 public:
  typedef std::shared_ptr<VoxgraphSubmap> Ptr;
  typedef std::shared_ptr<const VoxgraphSubmap> ConstPtr;
  enum class RegistrationPointType { kIsosurfacePoints = 0, kVoxels };

  struct TsdfMap {
    voxblox::Layer<voxblox::TsdfVoxel> layer;
    TsdfMap(float voxel_size, size_t vps) : layer(voxel_size, vps) {}
    const voxblox::Layer<voxblox::TsdfVoxel>& getTsdfLayer() const { return layer; }
  };
  struct EsdfMap {
    voxblox::Layer<voxblox::EsdfVoxel> layer;
    EsdfMap(float voxel_size, size_t vps) : layer(voxel_size, vps) {}
    const voxblox::Layer<voxblox::EsdfVoxel>& getEsdfLayer() const { return layer; }
  };

  VoxgraphSubmap(unsigned int id, const voxblox::Transformation& pose, float voxel_size, size_t vps)
      : id_(id), pose_(pose), tsdf_map_(voxel_size, vps), esdf_map_(voxel_size, vps) {}

  unsigned int getID() const { return id_; }
  const voxblox::Transformation& getPose() const { return pose_; }
  const TsdfMap& getTsdfMap() const { return tsdf_map_; }
  const EsdfMap& getEsdfMap() const { return esdf_map_; }
  TsdfMap& mutableTsdfMap() { return tsdf_map_; }
  EsdfMap& mutableEsdfMap() { return esdf_map_; }
  const WeightedSampler<RegistrationPoint>& getRegistrationPoints(RegistrationPointType type) const {
    return type == RegistrationPointType::kVoxels ? relevant_voxels_ : isosurface_vertices_;
  }
  WeightedSampler<RegistrationPoint>& mutableRegistrationPoints(RegistrationPointType type) {
    return type == RegistrationPointType::kVoxels ? relevant_voxels_ : isosurface_vertices_;
  }

 private:
  unsigned int id_;
  voxblox::Transformation pose_;
  TsdfMap tsdf_map_;
  EsdfMap esdf_map_;
  WeightedSampler<RegistrationPoint> relevant_voxels_;
  WeightedSampler<RegistrationPoint> isosurface_vertices_;
};
}  // namespace voxgraph
#endif
