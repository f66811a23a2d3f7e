// C entry points around the REFERENCE's own classes, compiled from /root/reference:
//   voxgraph/src/backend/constraint/cost_functions/registration_cost_function.cpp
//   voxgraph/src/frontend/submap_collection/voxgraph_submap.cpp
//   voxgraph/src/frontend/submap_collection/bounding_box.cpp
// against the stand-in headers of oracle/ref_shims (see its README).  TEST INFRASTRUCTURE:
// built into oracle/_ref/libref_reg.so, used by tests/ and tests/golden/make_ref_golden.py only.
#include <algorithm>
#include <cstdint>
#include <iostream>
#include <map>
#include <memory>
#include <random>
#include <sstream>
#include <streambuf>
#include <vector>

#include <cblox/core/tsdf_esdf_submap.h>
#include <ceres/ceres.h>
#include <voxblox/interpolator/interpolator.h>

// the samplers are private members; tests replace their contents with arbitrary point sets
#define private public
#include "voxgraph/frontend/submap_collection/voxgraph_submap.h"
#undef private
#include "voxgraph/backend/constraint/cost_functions/registration_cost_function.h"
#include "voxgraph/backend/constraint/cost_functions/relative_pose_cost_function.h"

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
voxblox::Transformation pose_from(const double p[4]) {
  voxblox::Transformation::Vector6 v;
  v[0] = p[0];
  v[1] = p[1];
  v[2] = p[2];
  v[3] = 0;
  v[4] = 0;
  v[5] = p[3];
  return voxblox::Transformation::exp(v);
}
// finishSubmap() prints to std::cout (voxgraph_submap.cpp:92,96): silence it
struct CoutSilencer {
  std::streambuf* old;
  std::ostringstream sink;
  CoutSilencer() : old(std::cout.rdbuf(sink.rdbuf())) {}
  ~CoutSilencer() { std::cout.rdbuf(old); }
};
VoxgraphSubmap::RegistrationPointType ptype(int32_t t) {
  return static_cast<VoxgraphSubmap::RegistrationPointType>(t);
}
}  // namespace

extern "C" {

// blocks: n x int32[3]; voxel arrays: n * vps^3 in voxblox linear order x + vps*(y + vps*z).
// Runs the reference's VoxgraphSubmap(T_M_S, id, tsdf_layer) constructor, injects the ESDF voxels
// (cblox::generateEsdf is a no-op in the stand-in), then the reference's finishSubmap():
// OBBs, findRelevantVoxelIndices, findIsosurfaceVertices.
void* refreg_submap_create(uint32_t id, const double pose_xyz_yaw[4], float voxel_size, int32_t vps,
                           int32_t n_blocks, const int32_t* block_index, const float* tsdf_distance,
                           const float* tsdf_weight, const float* esdf_distance,
                           const uint8_t* esdf_observed, double min_voxel_weight,
                           double max_voxel_distance, int32_t use_esdf_distance) {
  voxblox::Layer<voxblox::TsdfVoxel> tsdf_layer(voxel_size, static_cast<size_t>(vps));
  const size_t vox = static_cast<size_t>(vps) * vps * vps;
  for (int32_t b = 0; b < n_blocks; ++b) {
    voxblox::BlockIndex idx;
    for (int a = 0; a < 3; ++a) idx[a] = block_index[3 * b + a];
    auto tb = tsdf_layer.allocateBlockPtrByIndex(idx);
    for (size_t i = 0; i < vox; ++i) {
      voxblox::TsdfVoxel& t = tb->getVoxelByLinearIndex(i);
      t.distance = tsdf_distance[b * vox + i];
      t.weight = tsdf_weight[b * vox + i];
    }
  }
  auto* h = new SubmapHandle;
  h->submap = std::make_shared<VoxgraphSubmap>(pose_from(pose_xyz_yaw), id, tsdf_layer);
  voxblox::Layer<voxblox::EsdfVoxel>* esdf_layer = h->submap->getEsdfMapPtr()->getEsdfLayerPtr();
  for (int32_t b = 0; b < n_blocks; ++b) {
    voxblox::BlockIndex idx;
    for (int a = 0; a < 3; ++a) idx[a] = block_index[3 * b + a];
    auto eb = esdf_layer->allocateBlockPtrByIndex(idx);
    for (size_t i = 0; i < vox; ++i) {
      voxblox::EsdfVoxel& e = eb->getVoxelByLinearIndex(i);
      e.distance = esdf_distance ? esdf_distance[b * vox + i] : 0.0f;
      e.observed = esdf_observed ? esdf_observed[b * vox + i] != 0 : false;
    }
  }
  VoxgraphSubmap::Config::RegistrationFilter filter;
  filter.min_voxel_weight = min_voxel_weight;
  filter.max_voxel_distance = max_voxel_distance;
  filter.use_esdf_distance = use_esdf_distance != 0;
  h->submap->setRegistrationFilterConfig(filter);
  CoutSilencer quiet;
  h->submap->finishSubmap();
  return h;
}

void refreg_submap_destroy(void* submap) { delete static_cast<SubmapHandle*>(submap); }

void refreg_submap_set_pose(void* submap, const double pose_xyz_yaw[4]) {
  static_cast<SubmapHandle*>(submap)->submap->setPose(pose_from(pose_xyz_yaw));
}

// Layer::getAllAllocatedBlocks order (the order findRelevantVoxelIndices walks the blocks in)
int32_t refreg_submap_block_order(void* submap, int32_t* out) {
  voxblox::BlockIndexList blocks;
  static_cast<SubmapHandle*>(submap)->submap->getTsdfMap().getTsdfLayer().getAllAllocatedBlocks(&blocks);
  if (out)
    for (size_t b = 0; b < blocks.size(); ++b)
      for (int a = 0; a < 3; ++a) out[3 * b + a] = blocks[b][a];
  return static_cast<int32_t>(blocks.size());
}

int64_t refreg_submap_num_points(void* submap, int32_t point_type) {
  return static_cast<int64_t>(
      static_cast<SubmapHandle*>(submap)->submap->getRegistrationPoints(ptype(point_type)).size());
}

void refreg_submap_get_points(void* submap, int32_t point_type, float* xyz, float* distance, float* weight) {
  const auto& sampler = static_cast<SubmapHandle*>(submap)->submap->getRegistrationPoints(ptype(point_type));
  for (size_t i = 0; i < sampler.size(); ++i) {
    const RegistrationPoint& p = sampler[static_cast<int>(i)];
    xyz[3 * i] = p.position.x();
    xyz[3 * i + 1] = p.position.y();
    xyz[3 * i + 2] = p.position.z();
    distance[i] = p.distance;
    weight[i] = p.weight;
  }
}

// replaces a sampler's contents the way voxgraph_submap.cpp:194-197 / :233-235 fill it:
// addItem(RegistrationPoint{position, distance, weight}, weight)
void refreg_submap_set_points(void* submap, int32_t point_type, int64_t n, const float* xyz,
                              const float* distance, const float* weight) {
  VoxgraphSubmap& sm = *static_cast<SubmapHandle*>(submap)->submap;
  auto& sampler = ptype(point_type) == VoxgraphSubmap::RegistrationPointType::kVoxels
                      ? sm.relevant_voxels_
                      : sm.isosurface_vertices_;
  sampler.clear();
  for (int64_t i = 0; i < n; ++i) {
    RegistrationPoint p{voxblox::Point(xyz[3 * i], xyz[3 * i + 1], xyz[3 * i + 2]), distance[i], weight[i]};
    sampler.addItem(p, weight[i]);
  }
}

int32_t refreg_submap_isosurface_blocks(void* submap, int32_t* out) {
  const VoxgraphSubmap& sm = *static_cast<SubmapHandle*>(submap)->submap;
  int32_t n = 0;
  for (const voxblox::BlockIndex& b : sm.isosurface_blocks_) {
    if (out)
      for (int a = 0; a < 3; ++a) out[3 * n + a] = b[a];
    ++n;
  }
  return n;
}

// out = {min xyz, max xyz}
void refreg_submap_surface_obb(void* submap, float out[6]) {
  const voxgraph::BoundingBox box = static_cast<SubmapHandle*>(submap)->submap->getSubmapFrameSurfaceObb();
  for (int a = 0; a < 3; ++a) {
    out[a] = box.min[a];
    out[3 + a] = box.max[a];
  }
}
void refreg_submap_mission_surface_aabb(void* submap, float out[6]) {
  const voxgraph::BoundingBox box = static_cast<SubmapHandle*>(submap)->submap->getMissionFrameSurfaceAabb();
  for (int a = 0; a < 3; ++a) {
    out[a] = box.min[a];
    out[3 + a] = box.max[a];
  }
}
int32_t refreg_submap_overlaps_with(void* submap, void* other) {
  return static_cast<SubmapHandle*>(submap)->submap->overlapsWith(*static_cast<SubmapHandle*>(other)->submap) ? 1 : 0;
}

void* refreg_cost_create(void* reference_submap, void* reading_submap, int32_t point_type,
                         float sampling_ratio, double no_correspondence_cost, int32_t use_esdf_distance) {
  auto* c = new CostHandle;
  c->reference = static_cast<SubmapHandle*>(reference_submap)->submap;
  c->reading = static_cast<SubmapHandle*>(reading_submap)->submap;
  RegistrationCostFunction::Config cfg;
  cfg.registration_point_type = ptype(point_type);
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


// voxgraph::RelativePoseCostFunction (odometry and loop-closure edges,
// relative_pose_cost_function{.h,_inl.h}, normalize_angle.h): residuals of the functor with
// T = double, created through the reference's own Create().  observed = {x, y, z, yaw} of the
// measured relative pose; sqrt_information = 4x4 row-major.  observed_out (nullable) receives the
// values the functor actually stores: float translation widened to double, float yaw from log().
int32_t refreg_relative_pose_residual(const double observed[4], const double* sqrt_information,
                                      const double pose_a[4], const double pose_b[4],
                                      double residuals[4], double observed_out[4]) {
  const voxblox::Transformation T_obs = pose_from(observed);
  voxgraph::Constraint::InformationMatrix info;
  for (int r = 0; r < 4; ++r)
    for (int c = 0; c < 4; ++c) info(r, c) = sqrt_information[4 * r + c];
  if (observed_out) {
    for (int a = 0; a < 3; ++a) observed_out[a] = static_cast<double>(T_obs.getPosition()[a]);
    observed_out[3] = static_cast<double>(T_obs.log()[5]);
  }
  std::unique_ptr<ceres::CostFunction> cost(voxgraph::RelativePoseCostFunction::Create(T_obs, info));
  const double* parameters[2] = {pose_a, pose_b};
  return cost->Evaluate(parameters, residuals, nullptr) ? 1 : 0;
}

}  // extern "C"
