// Host-side mirror of voxgraph::RegistrationCostFunction over the C ABI.
//
// Drop-in for the class declared at
//   voxgraph/include/voxgraph/backend/constraint/cost_functions/registration_cost_function.h:11-83
// : same base class (ceres::CostFunction), same parameter blocks (two blocks of 4:
// {x,y,z,yaw} of the reference and of the reading submap, registration_cost_function.cpp:38-42),
// same num_residuals rule (:45-55), same Evaluate contract (:58-298, returns false
// when the summed reference weight is 0, :273).  Only this header needs Ceres; the
// library itself (libvoxgraph_amd.so) has no Ceres/Eigen/voxblox dependency.
//
// Usage at the reference's construction site (registration_constraint.cpp:33-35):
//   cost_function = new voxgraph_amd::GpuRegistrationCostFunction(
//       ctx, gpu_submap(first_submap_id), gpu_submap(second_submap_id), config);
// see INTEGRATION.md for the VoxgraphSubmap -> vgx_submap upload glue.
#ifndef VOXGRAPH_AMD_CPP_GPU_REGISTRATION_COST_FUNCTION_H_
#define VOXGRAPH_AMD_CPP_GPU_REGISTRATION_COST_FUNCTION_H_

#include <ceres/ceres.h>

#include <stdexcept>
#include <string>

#include "voxgraph_amd.h"

namespace voxgraph_amd {

class GpuRegistrationCostFunction : public ceres::CostFunction {
 public:
  // Mirrors RegistrationCostFunction::Config (registration_cost_function.h:17-41).
  struct Config {
    // VoxgraphSubmap::RegistrationPointType: kIsosurfacePoints = 0, kVoxels = 1
    int registration_point_type = VGX_POINTS_ISOSURFACE;
    float sampling_ratio = -1;
    double no_correspondence_cost = 0;
    bool use_esdf_distance = true;
    // visualize_* flags of the reference are debug-only and ignored on the GPU path
  };

  GpuRegistrationCostFunction(vgx_ctx ctx, vgx_submap reference_submap,
                              vgx_submap reading_submap, const Config& config)
      : ctx_(ctx) {
    vgx_reg_config cfg;
    vgx_reg_config_default(&cfg);
    cfg.registration_point_type = config.registration_point_type;
    cfg.sampling_ratio = config.sampling_ratio;
    cfg.no_correspondence_cost = config.no_correspondence_cost;
    cfg.use_esdf_distance = config.use_esdf_distance ? 1 : 0;
    const int rc = vgx_reg_create(ctx, reference_submap, reading_submap, &cfg, &reg_);
    if (rc != VGX_OK) {
      // the reference CHECK-fails on programmer errors (registration_cost_function.cpp:29-36)
      throw std::runtime_error(std::string("vgx_reg_create: ") + vgx_last_error(ctx));
    }
    // registration_cost_function.cpp:38-55
    mutable_parameter_block_sizes()->clear();
    mutable_parameter_block_sizes()->push_back(4);
    mutable_parameter_block_sizes()->push_back(4);
    set_num_residuals(static_cast<int>(vgx_reg_num_residuals(reg_)));
  }

  ~GpuRegistrationCostFunction() override { vgx_reg_destroy(reg_); }

  GpuRegistrationCostFunction(const GpuRegistrationCostFunction&) = delete;
  GpuRegistrationCostFunction& operator=(const GpuRegistrationCostFunction&) = delete;

  // registration_cost_function.h:47-48
  bool Evaluate(double const* const* parameters, double* residuals,
                double** jacobians) const override {
    double* jac_ref = jacobians ? jacobians[0] : nullptr;
    double* jac_read = jacobians ? jacobians[1] : nullptr;
    const int rc = vgx_reg_evaluate(reg_, parameters[0], parameters[1], residuals, jac_ref, jac_read);
    if (rc == VGX_OK) return true;
    if (rc == VGX_EVALUATE_FALSE) return false;  // summed_reference_weight == 0 (.cpp:273)
    throw std::runtime_error(std::string("vgx_reg_evaluate: ") + vgx_last_error(ctx_));
  }

  vgx_reg handle() const { return reg_; }

 private:
  vgx_ctx ctx_;
  vgx_reg reg_ = nullptr;
};

}  // namespace voxgraph_amd

#endif  // VOXGRAPH_AMD_CPP_GPU_REGISTRATION_COST_FUNCTION_H_
