// Only EsdfIntegrator::Config is named by the translation units compiled here; ESDF generation
// itself (voxblox) is restated separately in oracle/esdf_oracle.c.  TEST INFRASTRUCTURE.
#ifndef ORACLE_REF_SHIMS_VOXBLOX_INTEGRATOR_ESDF_INTEGRATOR_H_
#define ORACLE_REF_SHIMS_VOXBLOX_INTEGRATOR_ESDF_INTEGRATOR_H_
#include "voxblox/core/esdf_map.h"
#include "voxblox/core/tsdf_map.h"
namespace voxblox {
class EsdfIntegrator {
 public:
  struct Config {
    FloatingPoint max_distance_m = 2.0;
    FloatingPoint default_distance_m = 2.0;
    FloatingPoint min_distance_m = 0.2;
  };
};
}  // namespace voxblox
#endif
