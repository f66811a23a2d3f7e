// Stand-in for cblox::TsdfEsdfSubmap ([recalled]): pose, id, a TSDF map and an ESDF map of the
// same geometry.  generateEsdf() is a no-op here: the driver injects the ESDF voxels (ESDF
// generation is voxblox's EsdfIntegrator, restated separately in oracle/esdf_oracle.c).
// TEST INFRASTRUCTURE -- see oracle/ref_shims/README.md.
#ifndef ORACLE_REF_SHIMS_CBLOX_CORE_TSDF_ESDF_SUBMAP_H_
#define ORACLE_REF_SHIMS_CBLOX_CORE_TSDF_ESDF_SUBMAP_H_
#include <istream>
#include <memory>
#include <utility>

#include "voxblox/core/esdf_map.h"
#include "voxblox/core/tsdf_map.h"
#include "voxblox/integrator/esdf_integrator.h"
#include "voxblox/interpolator/interpolator.h"
#include "cblox/core/common.h"
namespace cblox {

class TsdfEsdfSubmap {
 public:
  typedef std::shared_ptr<TsdfEsdfSubmap> Ptr;
  struct Config : voxblox::TsdfMap::Config, voxblox::EsdfMap::Config {};

  TsdfEsdfSubmap(const Transformation& T_M_S, SubmapID submap_id, Config config,
                 voxblox::EsdfIntegrator::Config = voxblox::EsdfIntegrator::Config())
      : submap_id_(submap_id), T_M_S_(T_M_S), mapping_interval_(0, 0) {
    tsdf_map_ = std::make_shared<voxblox::TsdfMap>(config);
    esdf_map_ = std::make_shared<voxblox::EsdfMap>(config);
  }
  virtual ~TsdfEsdfSubmap() {}
  virtual void finishSubmap() {}
  void generateEsdf() {}

  const Transformation& getPose() const { return T_M_S_; }
  void setPose(const Transformation& T_M_S) { T_M_S_ = T_M_S; }
  SubmapID getID() const { return submap_id_; }
  const voxblox::TsdfMap& getTsdfMap() const { return *tsdf_map_; }
  const voxblox::EsdfMap& getEsdfMap() const { return *esdf_map_; }
  voxblox::TsdfMap::Ptr getTsdfMapPtr() { return tsdf_map_; }
  voxblox::EsdfMap::Ptr getEsdfMapPtr() { return esdf_map_; }
  voxblox::FloatingPoint block_size() const { return tsdf_map_->block_size(); }

  static Ptr LoadFromStream(const Config&, std::istream*, uint64_t*) { return nullptr; }

 protected:
  SubmapID submap_id_;
  Transformation T_M_S_;
  voxblox::TsdfMap::Ptr tsdf_map_;
  voxblox::EsdfMap::Ptr esdf_map_;
  std::pair<int64_t, int64_t> mapping_interval_;
};
}  // namespace cblox
#endif
