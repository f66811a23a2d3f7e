// C entry points around the REFERENCE's voxgraph::RegistrationCostFunction, compiled from
// /root/reference/voxgraph/src/backend/constraint/cost_functions/registration_cost_function.cpp
// against the stand-in headers of oracle/ref_shims (see its README).  TEST INFRASTRUCTURE:
// built into oracle/_ref/libref_reg.so, used by tests/ and tests/golden/make_ref_golden.py only.
#include <cstdint>
#include <memory>

#include "voxgraph/backend/constraint/cost_functions/registration_cost_function.h"

using voxgraph::RegistrationCostFunction;
using voxgraph::RegistrationPoint;
using voxgraph::VoxgraphSubmap;

namespace {
struct SubmapHandle {
  std::shared_ptr<VoxgraphSubmap> submap;
};
struct CostHandle {
  std::shared_ptr<VoxgraphSubmap> reference, reading;  // must outlive the cost function
  std::unique_ptr<RegistrationCostFunction> cost;
};
}  // namespace

extern "C" {

// blocks: n x int32[3]; voxel arrays: n * vps^3 in voxblox linear order x + vps*(y + vps*z)
void* refreg_submap_create(uint32_t id, const double pose_xyz_yaw[4], float voxel_size, int32_t vps,
                           int32_t n_blocks, const int32_t* block_index, const float* tsdf_distance,
                           const float* tsdf_weight, const float* esdf_distance,
                           const uint8_t* esdf_observed) {
  voxblox::Transformation::Vector6 v;
  v[0] = pose_xyz_yaw[0];
  v[1] = pose_xyz_yaw[1];
  v[2] = pose_xyz_yaw[2];
  v[3] = 0;
  v[4] = 0;
  v[5] = pose_xyz_yaw[3];
  auto* h = new SubmapHandle;
  h->submap = std::make_shared<VoxgraphSubmap>(id, voxblox::Transformation::exp(v), voxel_size,
                                               static_cast<size_t>(vps));
  const size_t vox = static_cast<size_t>(vps) * vps * vps;
  for (int32_t b = 0; b < n_blocks; ++b) {
    voxblox::BlockIndex idx;
    idx[0] = block_index[3 * b + 0];
    idx[1] = block_index[3 * b + 1];
    idx[2] = block_index[3 * b + 2];
    auto tb = h->submap->mutableTsdfMap().layer.allocateBlockPtrByIndex(idx);
    auto eb = h->submap->mutableEsdfMap().layer.allocateBlockPtrByIndex(idx);
    for (size_t i = 0; i < vox; ++i) {
      voxblox::TsdfVoxel& t = tb->getVoxelByLinearIndex(i);
      t.distance = tsdf_distance[b * vox + i];
      t.weight = tsdf_weight[b * vox + i];
      voxblox::EsdfVoxel& e = eb->getVoxelByLinearIndex(i);
      e.distance = esdf_distance ? esdf_distance[b * vox + i] : 0.0f;
      e.observed = esdf_observed ? esdf_observed[b * vox + i] != 0 : false;
    }
  }
  return h;
}

void refreg_submap_destroy(void* submap) { delete static_cast<SubmapHandle*>(submap); }

// appends registration points exactly as voxgraph_submap.cpp:194-197 / :233-235 do:
// addItem(RegistrationPoint{position, distance, weight}, weight)
void refreg_submap_add_points(void* submap, int32_t point_type, int64_t n, const float* xyz,
                              const float* distance, const float* weight) {
  auto* h = static_cast<SubmapHandle*>(submap);
  auto& sampler = h->submap->mutableRegistrationPoints(
      static_cast<VoxgraphSubmap::RegistrationPointType>(point_type));
  for (int64_t i = 0; i < n; ++i) {
    RegistrationPoint p{voxblox::Point(xyz[3 * i], xyz[3 * i + 1], xyz[3 * i + 2]), distance[i], weight[i]};
    sampler.addItem(p, weight[i]);
  }
}

void* refreg_cost_create(void* reference_submap, void* reading_submap, int32_t point_type,
                         float sampling_ratio, double no_correspondence_cost, int32_t use_esdf_distance) {
  auto* c = new CostHandle;
  c->reference = static_cast<SubmapHandle*>(reference_submap)->submap;
  c->reading = static_cast<SubmapHandle*>(reading_submap)->submap;
  RegistrationCostFunction::Config cfg;
  cfg.registration_point_type = static_cast<VoxgraphSubmap::RegistrationPointType>(point_type);
  cfg.sampling_ratio = sampling_ratio;
  cfg.no_correspondence_cost = no_correspondence_cost;
  cfg.use_esdf_distance = use_esdf_distance != 0;
  c->cost.reset(new RegistrationCostFunction(c->reference, c->reading, cfg));
  return c;
}

void refreg_cost_destroy(void* cost) { delete static_cast<CostHandle*>(cost); }

int32_t refreg_cost_num_residuals(void* cost) { return static_cast<CostHandle*>(cost)->cost->num_residuals(); }

// ceres::CostFunction::Evaluate; jac_ref / jac_read may each be NULL, want_jacobians = 0 passes
// jacobians == nullptr.  Returns 1 for true, 0 for false.
int32_t refreg_cost_evaluate(void* cost, const double ref_pose[4], const double read_pose[4],
                             int32_t want_jacobians, double* residuals, double* jac_ref, double* jac_read) {
  const double* parameters[2] = {ref_pose, read_pose};
  double* jacobians[2] = {jac_ref, jac_read};
  return static_cast<CostHandle*>(cost)->cost->Evaluate(parameters, residuals,
                                                        want_jacobians ? jacobians : nullptr)
             ? 1
             : 0;
}

}  // extern "C"
