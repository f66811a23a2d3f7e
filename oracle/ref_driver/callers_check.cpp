// The reference's OWN callers on top of either cost function (TEST INFRASTRUCTURE; `make -C oracle ref` builds
// oracle/_ref/callers_check_reference and oracle/_ref/callers_check_gpu where /root/reference exists; run on the GPU
// box by tests/test_callers_gpu.py).
//
// Compiled FROM /root/reference, where they lie:
//     voxgraph/src/backend/pose_graph.cpp                      PoseGraph::addRegistrationConstraint / initialize / optimize
//     voxgraph/src/backend/constraint/constraint_collection.cpp
//     voxgraph/src/backend/constraint/registration_constraint.cpp   (*)
//     voxgraph/src/backend/node/node.cpp, node_collection.cpp, pose/pose_4d.cpp
//     voxgraph/src/tools/submap_registration_helper.cpp             (*)
//   + registration_cost_function.cpp, voxgraph_submap.cpp, bounding_box.cpp (as for libref_reg.so)
// against the stand-in headers of oracle/ref_shims and the Ceres stand-in of tests/stubs (a dense Levenberg-Marquardt
// with Ceres' acceptance rule; the real Ceres is absent from this image).
// (*) in the _gpu binary these two are compiled from copies made at build time by ONE sed edit each -- the edit
// INTEGRATION.md section 3 shows: `new RegistrationCostFunction(` -> `voxgraph_amd::MakeGpuRegistrationCostFunction(`
// (voxgraph_amd/cpp/gpu_submap_registry.h) -- no reference source is committed.
// NOT compiled from the reference (stated, not hidden): constraint.cpp (Eigen's LLT / LDLT on dynamic matrices) and the
// odometry / loop-closure constraints' .cpp (AutoDiff over Eigen-of-Jet types): the base-class constructor below does the
// same Cholesky on the fixed 4 x 4 matrix, the two addToProblem() abort -- this graph holds registration constraints only.
//
// Both binaries build the same four-submap graph (first submap constant, pose_graph_interface.cpp:30-32), perturb the
// other three poses, call PoseGraph::optimize() and SubmapRegistrationHelper::testRegistration(), and print the poses:
// tests/test_callers_gpu.py requires them within 1 mm / 0.01 deg of each other (north_star's solve tolerance).
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <iostream>
#include <memory>
#include <sstream>
#include <vector>

#include <cblox/core/submap_collection.h>
#include <ceres/ceres.h>

#include "voxgraph/backend/pose_graph.h"
#include "voxgraph/frontend/submap_collection/voxgraph_submap.h"
#include "voxgraph/tools/submap_registration_helper.h"

#ifdef VGX_CALLERS_GPU
#include "gpu_submap_registry.h"
#endif

using voxgraph::VoxgraphSubmap;

// ---- what is not compiled from the reference (see the header comment) ------------------------------------------------
namespace voxgraph {
Constraint::Constraint(Constraint::ConstraintId constraint_id, const Constraint::Config& config)
    : constraint_id_(constraint_id) {
  // constraint.cpp:8-16: the lower Cholesky factor of the information matrix (allow_semi_definite... not supported here)
  CHECK(!config.allow_semi_definite_information_matrix);
  InformationMatrix L;
  L.setZero();
  for (int j = 0; j < 4; ++j) {
    double s = config.information_matrix(j, j);
    for (int k = 0; k < j; ++k) s -= L(j, k) * L(j, k);
    CHECK(s > 0) << "The square root of the information matrix could not be computed";
    L(j, j) = std::sqrt(s);
    for (int i = j + 1; i < 4; ++i) {
      double t = config.information_matrix(i, j);
      for (int k = 0; k < j; ++k) t -= L(i, k) * L(j, k);
      L(i, j) = t / L(j, j);
    }
  }
  sqrt_information_matrix_ = L;
}
void RelativePoseConstraint::addToProblem(const NodeCollection&, ceres::Problem*) {
  LOG(FATAL) << "callers_check: relative pose constraints are not part of this check";
}
void AbsolutePoseConstraint::addToProblem(const NodeCollection&, ceres::Problem*) {
  LOG(FATAL) << "callers_check: absolute pose constraints are not part of this check";
}
}  // namespace voxgraph

namespace {
// analytic scene in the mission frame: ground, a sphere, two boxes along the track
float scene_sdf(float x, float y, float z) {
  auto box = [](float px, float py, float pz, float cx, float cy, float cz, float hx, float hy, float hz) {
    const float bx = std::fabs(px - cx) - hx, by = std::fabs(py - cy) - hy, bz = std::fabs(pz - cz) - hz;
    const float ox = std::fmax(bx, 0.0f), oy = std::fmax(by, 0.0f), oz = std::fmax(bz, 0.0f);
    return std::sqrt(ox * ox + oy * oy + oz * oz) + std::fmin(std::fmax(bx, std::fmax(by, bz)), 0.0f);
  };
  const float sphere = std::sqrt((x - 2.1f) * (x - 2.1f) + (y - 1.5f) * (y - 1.5f) + (z - 1.2f) * (z - 1.2f)) - 0.8f;
  const float ground = z - 0.35f;
  const float b1 = box(x, y, z, 0.9f, 2.4f, 0.6f, 0.35f, 0.5f, 0.4f), b2 = box(x, y, z, 3.6f, 0.9f, 0.8f, 0.45f, 0.3f, 0.6f);
  const float b3 = box(x, y, z, 5.0f, 2.2f, 0.7f, 0.3f, 0.55f, 0.5f);
  return std::fmin(std::fmin(std::fmin(sphere, ground), std::fmin(b1, b2)), b3);
}

voxblox::Transformation pose_of(const double p[4]) {
  voxblox::Transformation::Vector6 v;
  v[0] = p[0]; v[1] = p[1]; v[2] = p[2]; v[3] = 0; v[4] = 0; v[5] = p[3];
  return voxblox::Transformation::exp(v);
}

std::shared_ptr<VoxgraphSubmap> make_submap(unsigned id, const double pose[4], float voxel_size, int vps, int blocks_per_side,
                                            float trunc, float esdf_max) {
  const voxblox::Transformation T_M_S = pose_of(pose);
  voxblox::Layer<voxblox::TsdfVoxel> tsdf(voxel_size, vps);
  for (int bx = 0; bx < blocks_per_side; ++bx)
    for (int by = 0; by < blocks_per_side; ++by)
      for (int bz = 0; bz < blocks_per_side; ++bz) {
        voxblox::BlockIndex idx;
        idx[0] = bx; idx[1] = by; idx[2] = bz;
        auto block = tsdf.allocateBlockPtrByIndex(idx);
        for (size_t i = 0; i < block->num_voxels(); ++i) {
          const voxblox::Point p_m = T_M_S * block->computeCoordinatesFromLinearIndex(i);
          const float d = scene_sdf(p_m.x(), p_m.y(), p_m.z());
          voxblox::TsdfVoxel& t = block->getVoxelByLinearIndex(i);
          t.distance = std::fmax(-trunc, std::fmin(trunc, d));
          t.weight = std::fabs(d) < 2.0f * trunc ? 10.0f : 0.0f;
        }
      }
  auto submap = std::make_shared<VoxgraphSubmap>(T_M_S, id, tsdf);
  voxblox::Layer<voxblox::EsdfVoxel>* esdf = submap->getEsdfMapPtr()->getEsdfLayerPtr();
  voxblox::BlockIndexList list;
  tsdf.getAllAllocatedBlocks(&list);
  for (const voxblox::BlockIndex& idx : list) {
    auto block = esdf->allocateBlockPtrByIndex(idx);
    for (size_t i = 0; i < block->num_voxels(); ++i) {
      const voxblox::Point p_m = T_M_S * block->computeCoordinatesFromLinearIndex(i);
      const float d = scene_sdf(p_m.x(), p_m.y(), p_m.z());
      voxblox::EsdfVoxel& e = block->getVoxelByLinearIndex(i);
      e.distance = std::fmax(-esdf_max, std::fmin(esdf_max, d));
      e.observed = std::fabs(d) < esdf_max;
    }
  }
  std::ostringstream sink;  // finishSubmap() prints point counts to std::cout
  std::streambuf* old = std::cout.rdbuf(sink.rdbuf());
  submap->finishSubmap();
  std::cout.rdbuf(old);
  return submap;
}
}  // namespace

int main() {
#ifdef VGX_CALLERS_GPU
  vgx_ctx ctx = nullptr;
  if (vgx_ctx_create(0, &ctx) != VGX_OK) {
    std::fprintf(stderr, "no device: %s\n", vgx_last_error(nullptr));
    return 2;
  }
  voxgraph_amd::GpuSubmapRegistry::instance().setContext(ctx);
#endif
  // four submaps along the track, each 3.2 m cubed, overlapping their neighbours by more than half
  const double truth[4][4] = {{0.0, 0.0, 0.0, 0.0}, {1.3, 0.2, 0.02, 0.06}, {2.5, -0.1, 0.0, -0.05}, {3.4, 0.15, -0.03, 0.04}};
  const double drift[4][4] = {{0, 0, 0, 0}, {0.06, -0.04, 0.03, 0.015}, {-0.05, 0.07, -0.02, -0.02}, {0.08, 0.03, 0.04, 0.01}};
  std::vector<std::shared_ptr<VoxgraphSubmap>> submaps;
  auto collection = std::make_shared<cblox::SubmapCollection<VoxgraphSubmap>>();
  for (unsigned k = 0; k < 4; ++k) {
    submaps.push_back(make_submap(10 + k, truth[k], 0.1f, 16, 2, 0.3f, 1.0f));
    collection->addSubmap(submaps.back());
  }
  int rc = 0;
  for (int point_type = 1; point_type >= 0; --point_type) {   // kVoxels, then kIsosurfacePoints (mirrored: pose_graph.cpp:62-71)
    voxgraph::PoseGraph graph;
    for (unsigned k = 0; k < 4; ++k) {
      voxgraph::SubmapNode::Config node;
      node.submap_id = 10 + k;
      node.set_constant = k == 0;   // pose_graph_interface.cpp:30-32
      double start[4];
      for (int a = 0; a < 4; ++a) start[a] = truth[k][a] + drift[k][a];
      node.T_mission_node_initial = pose_of(start);
      graph.addSubmapNode(node);
    }
    for (unsigned a = 0; a < 4; ++a)
      for (unsigned b = a + 1; b < 4 && b <= a + 2; ++b) {
        voxgraph::RegistrationConstraint::Config c;
        c.first_submap_id = 10 + a;
        c.second_submap_id = 10 + b;
        c.first_submap_ptr = submaps[a];
        c.second_submap_ptr = submaps[b];
        c.information_matrix.setIdentity();
        c.registration.registration_point_type = static_cast<VoxgraphSubmap::RegistrationPointType>(point_type);
        c.registration.sampling_ratio = -1;
        graph.addRegistrationConstraint(c);
      }
    std::ostringstream sink;  // optimize() prints the solver report
    std::streambuf* old = std::cout.rdbuf(sink.rdbuf());
    graph.optimize();
    std::cout.rdbuf(old);
    const auto& summary = graph.getSolverSummaries().back();
    std::printf("SOLVE point_type=%d iterations=%d initial_cost=%.9e final_cost=%.9e %s\n", point_type, summary.num_iterations,
                summary.initial_cost, summary.final_cost, summary.termination);
    if (!(summary.final_cost < 0.2 * summary.initial_cost)) rc = 1;
    for (const auto& kv : graph.getSubmapPoses()) {
      const voxblox::Transformation::Vector6 v = kv.second.log();
      std::printf("POSE point_type=%d submap=%u %.9f %.9f %.9f %.9f\n", point_type, kv.first, (double)v[0], (double)v[1], (double)v[2],
                  (double)v[5]);
    }
    // PoseGraph::getVisualizationEdges (pose_graph.cpp:167-211): one edge per residual block, sum of squared residuals
    double edge_sum = 0;
    for (const auto& e : graph.getVisualizationEdges()) edge_sum += e.residual;
    std::printf("EDGES point_type=%d sum_sq_residuals=%.9e\n", point_type, edge_sum);
  }
  // SubmapRegistrationHelper::testRegistration (submap_registration_helper.cpp:15-71): reading submap 1 against submap 0
  {
    voxgraph::SubmapRegistrationHelper::Options opt;
    opt.registration.registration_point_type = VoxgraphSubmap::RegistrationPointType::kVoxels;
    opt.solver.max_num_iterations = 30;
    voxgraph::SubmapRegistrationHelper helper(collection, opt);
    double reading[4];
    for (int a = 0; a < 4; ++a) reading[a] = truth[1][a] + drift[1][a];
    ceres::Solver::Summary summary;
    const bool usable = helper.testRegistration(10, 11, reading, &summary);
    std::printf("HELPER usable=%d iterations=%d final_cost=%.9e pose %.9f %.9f %.9f %.9f\n", (int)usable, summary.num_iterations,
                summary.final_cost, reading[0], reading[1], reading[2], reading[3]);
    if (!usable) rc = 1;
  }
#ifdef VGX_CALLERS_GPU
  voxgraph_amd::GpuSubmapRegistry::instance().clear();
  vgx_ctx_destroy(ctx);
#endif
  return rc;
}
