// REHEARSAL STAND-IN (oracle/pin_dryrun/README.md) for voxblox/integrator/tsdf_integrator.h: voxblox's integrator classes
// with their public shape [recalled] -- TsdfIntegratorBase::Config, FastTsdfIntegrator / MergedTsdfIntegrator(config,
// Layer<TsdfVoxel>*), integratePointCloud(T_G_C, points_C, colors, freespace_points) -- implemented by oracle/tsdf_oracle.c,
// so that `make -C oracle pin-dryrun` can compile and run ref_driver/voxblox_tsdf_pin.cpp.  TEST INFRASTRUCTURE; proves
// nothing about voxblox.
#ifndef ORACLE_PIN_DRYRUN_VOXBLOX_INTEGRATOR_TSDF_INTEGRATOR_H_
#define ORACLE_PIN_DRYRUN_VOXBLOX_INTEGRATOR_TSDF_INTEGRATOR_H_
#include <memory>
#include <string>
#include <vector>

#include "voxblox/core/common.h"
#include "voxblox/core/layer.h"
#include "voxblox/core/voxel.h"

struct orc_tsdf_layer;
struct orc_tsdf_integrator;

namespace voxblox {
typedef AlignedVector<Point> Pointcloud;
typedef AlignedVector<Color> Colors;

class TsdfIntegratorBase {
 public:
  typedef std::shared_ptr<TsdfIntegratorBase> Ptr;
  struct Config {
    float default_truncation_distance = 0.1f;
    float max_weight = 10000.0f;
    bool voxel_carving_enabled = true;
    float min_ray_length_m = 0.1f;
    float max_ray_length_m = 5.0f;
    bool use_const_weight = false;
    bool allow_clear = true;
    bool use_weight_dropoff = true;
    bool use_sparsity_compensation_factor = false;
    float sparsity_compensation_factor = 1.0f;
    size_t integrator_threads = 1;
    std::string integration_order_mode = "mixed";
    bool enable_anti_grazing = false;
    float start_voxel_subsampling_factor = 2.0f;
    int max_consecutive_ray_collisions = 2;
    int clear_checks_every_n_frames = 1;
  };
  TsdfIntegratorBase(const Config& config, Layer<TsdfVoxel>* layer, bool merged);
  virtual ~TsdfIntegratorBase();
  void integratePointCloud(const Transformation& T_G_C, const Pointcloud& points_C, const Colors& colors,
                           const bool freespace_points = false);
  void setLayer(Layer<TsdfVoxel>* layer) { layer_ = layer; }

 protected:
  Config config_;
  Layer<TsdfVoxel>* layer_;
  bool merged_;
  orc_tsdf_layer* orc_layer_;
  orc_tsdf_integrator* orc_;
};

class FastTsdfIntegrator : public TsdfIntegratorBase {
 public:
  FastTsdfIntegrator(const Config& config, Layer<TsdfVoxel>* layer) : TsdfIntegratorBase(config, layer, false) {}
};
class MergedTsdfIntegrator : public TsdfIntegratorBase {
 public:
  MergedTsdfIntegrator(const Config& config, Layer<TsdfVoxel>* layer) : TsdfIntegratorBase(config, layer, true) {}
};
}  // namespace voxblox
#endif
