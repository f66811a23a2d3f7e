// The reference's OWN callers on top of either cost function (TEST INFRASTRUCTURE; `make -C oracle ref` builds
// oracle/_ref/callers_check_reference and oracle/_ref/callers_check_gpu where /root/reference exists; run on the GPU
// box by tests/test_callers_gpu.py).
//
// Compiled FROM /root/reference, where they lie:
//     voxgraph/src/backend/pose_graph.cpp                      PoseGraph::addRegistrationConstraint / initialize / optimize
//     voxgraph/src/backend/constraint/constraint_collection.cpp
//     voxgraph/src/backend/constraint/constraint.cpp                (round 6: Eigen::LLT / LDLT<MatrixXd> stand-ins)
//     voxgraph/src/backend/constraint/relative_pose_constraint.cpp  (round 6: ceres::Jet + AutoDiffCostFunction stand-ins,
//     voxgraph/src/backend/constraint/absolute_pose_constraint.cpp   relative_pose_cost_function_inl.h differentiated by them)
//     voxgraph/src/backend/constraint/registration_constraint.cpp   (*)
//     voxgraph/src/backend/node/node.cpp, node_collection.cpp, pose/pose_4d.cpp
//     voxgraph/src/tools/submap_registration_helper.cpp             (*)
//   + registration_cost_function.cpp, voxgraph_submap.cpp, bounding_box.cpp (as for libref_reg.so)
// against the stand-in headers of oracle/ref_shims and the Ceres stand-in of tests/stubs (a dense Levenberg-Marquardt
// with Ceres' acceptance rule; the real Ceres is absent from this image).
// (*) in the _gpu binary these two are compiled from copies made at build time by ONE sed edit each -- the edit
// INTEGRATION.md section 3 shows: `new RegistrationCostFunction(` -> `voxgraph_amd::MakeGpuRegistrationCostFunction(`
// (voxgraph_amd/cpp/gpu_submap_registry.h) -- no reference source is committed.
// Nothing of the reference's backend is hand-written here any more (round 5 carried its own Cholesky and two aborting
// addToProblem stubs).  NOT compiled: voxgraph/src/tools/evaluation/map_evaluation.cpp (the third construction site of
// the cost function, :143-144) -- the file is a ROS node (publishers, voxblox message conversions); its alignment
// problem (:116-161) is RESTATED below as a scenario, with the same options, and says so.
//
// Both binaries build the same four-submap graph as voxgraph builds it: first submap constant
// (pose_graph_interface.cpp:30-32), registration constraints between overlapping submaps, ODOMETRY edges between
// consecutive submaps with the shipped information matrix (voxgraph_mapper.yaml:41-47: 1, 1, 2500, 2500; the measured
// relative pose carries a small error, as odometry does), and a HEIGHT measurement on the last submap (absolute pose
// against the mission frame, semi-definite information: measurement_templates.cpp:27-29 -- the LDLT branch of
// constraint.cpp).  They perturb the poses, call PoseGraph::optimize(), getVisualizationEdges(), getEdgeCovarianceMap(), the
// two-stage optimize(true) / optimize() of pose_graph_interface.cpp:182-191 and SubmapRegistrationHelper::testRegistration(),
// run the map-evaluation alignment, and print the poses and covariance blocks: tests/test_callers_gpu.py requires them within 1 mm / 0.01 deg
// of each other (north_star's solve tolerance).
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

// The four-submap graph as voxgraph builds it (see the header comment), poses starting from truth + drift
void fill_graph(voxgraph::PoseGraph& graph, const std::vector<std::shared_ptr<VoxgraphSubmap>>& submaps, int point_type,
                const double truth[4][4], const double drift[4][4]) {
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
  // odometry edges between consecutive submaps (pose_graph_interface.cpp:41-66), shipped information matrix
  for (unsigned a = 0; a + 1 < 4; ++a) {
    voxgraph::RelativePoseConstraint::Config c;
    c.origin_submap_id = 10 + a;
    c.destination_submap_id = 11 + a;
    double measured[4];   // the true relative pose seen through a little odometry error
    const voxblox::Transformation T_ab = pose_of(truth[a]).inverse() * pose_of(truth[a + 1]);
    const voxblox::Transformation::Vector6 rel = T_ab.log();
    measured[0] = rel[0] + 0.004 * (a + 1); measured[1] = rel[1] - 0.003; measured[2] = rel[2] + 0.001; measured[3] = rel[5] + 0.001;
    c.T_origin_destination = pose_of(measured);
    c.information_matrix.setZero();
    c.information_matrix(0, 0) = 1.0; c.information_matrix(1, 1) = 1.0;
    c.information_matrix(2, 2) = 2500.0; c.information_matrix(3, 3) = 2500.0;
    graph.addRelativePoseConstraint(c);
  }
  // a height measurement on the last submap: z only, against the mission frame (semi-definite: constraint.cpp's LDLT branch)
  {
    voxgraph::ReferenceFrameNode::Config frame;
    frame.reference_frame_id = 0;
    frame.set_constant = true;
    frame.T_mission_node_initial = voxblox::Transformation();
    graph.addReferenceFrameNode(frame);
    voxgraph::AbsolutePoseConstraint::Config c;
    c.reference_frame_id = 0;
    c.submap_id = 13;
    double at[4] = {0.0, 0.0, truth[3][2] + 0.001, 0.0};
    c.T_ref_submap = pose_of(at);
    c.information_matrix.setZero();
    c.information_matrix(2, 2) = 2500.0;
    c.allow_semi_definite_information_matrix = true;
    graph.addAbsolutePoseConstraint(c);
  }
}

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
    fill_graph(graph, submaps, point_type, truth, drift);
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
    // PoseGraph::getEdgeCovarianceMap (pose_graph.cpp:117-163; asked for by loop_closure_edge_server.cpp:46 through
    // pose_graph_interface.cpp:207-218 for every overlapping pair): ceres::Covariance over the problem optimize() left,
    // i.e. one more evaluation of every cost function WITH Jacobians at the final poses
    {
      voxgraph::PoseGraph::EdgeCovarianceMap cov;
      for (unsigned a = 0; a < 4; ++a)
        for (unsigned b = a + 1; b < 4 && b <= a + 2; ++b)
          cov.emplace(voxgraph::SubmapIdPair(10 + a, 10 + b), voxgraph::PoseGraph::EdgeCovarianceMatrix::Zero());
      const bool ok = graph.getEdgeCovarianceMap(&cov);
      std::printf("COVARIANCE point_type=%d ok=%d pairs=%zu\n", point_type, (int)ok, cov.size());
      if (!ok) rc = 1;
      for (const auto& kv : cov) {
        std::printf("COV point_type=%d pair=%u,%u", point_type, kv.first.first, kv.first.second);
        for (int i = 0; i < 4; ++i)
          for (int j = 0; j < 4; ++j) std::printf(" %.12e", kv.second(i, j));
        std::printf("\n");
      }
    }
    // The two-stage optimisation after a loop closure (pose_graph_interface.cpp:182-191): first WITHOUT the registration
    // constraints (optimize(true): constraint_collection.cpp skips them), then with all of them.  The loop closure ties the
    // last submap to the first (pose_graph_interface.cpp:68-92); a fresh graph, poses starting from the drift again.
    if (point_type == 1) {
      voxgraph::PoseGraph graph;
      fill_graph(graph, submaps, point_type, truth, drift);
      voxgraph::RelativePoseConstraint::Config c;
      c.origin_submap_id = 10;
      c.destination_submap_id = 13;
      const voxblox::Transformation T_ab = pose_of(truth[0]).inverse() * pose_of(truth[3]);
      const voxblox::Transformation::Vector6 rel = T_ab.log();
      double measured[4] = {rel[0] - 0.002, rel[1] + 0.002, rel[2], rel[5] - 0.0005};
      c.T_origin_destination = pose_of(measured);
      c.information_matrix.setIdentity();
      graph.addRelativePoseConstraint(c);
      std::ostringstream sink2;
      std::streambuf* old2 = std::cout.rdbuf(sink2.rdbuf());
      graph.optimize(true);
      const auto first = graph.getSolverSummaries().back();
      graph.optimize();
      std::cout.rdbuf(old2);
      const auto second = graph.getSolverSummaries().back();
      std::printf("TWOSTAGE iterations=%d,%d final_cost=%.9e,%.9e\n", first.num_iterations, second.num_iterations, first.final_cost,
                  second.final_cost);
      for (const auto& kv : graph.getSubmapPoses()) {
        const voxblox::Transformation::Vector6 v = kv.second.log();
        std::printf("POSE2 submap=%u %.9f %.9f %.9f %.9f\n", kv.first, (double)v[0], (double)v[1], (double)v[2], (double)v[5]);
      }
    }
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
  // MapEvaluation::alignSubmapAtoSubmapB (map_evaluation.cpp:116-161), RESTATED (the file itself is a ROS node): submap A
  // = a copy of submap 1 whose pose the solver is free to move, against submap B = submap 0 held constant; kVoxels, ESDF
  // distance, no sampling, 200 iterations, parameter_tolerance 1e-12; the cost function built from (submap_B, submap_A, config)
  {
    ceres::Problem problem;
    ceres::Solver::Summary summary;
    ceres::Solver::Options ceres_options;
    ceres_options.max_num_iterations = 200;
    ceres_options.parameter_tolerance = 1e-12;
    voxgraph::RegistrationCostFunction::Config cost_config;
    cost_config.use_esdf_distance = true;
    cost_config.sampling_ratio = -1;
    cost_config.registration_point_type = VoxgraphSubmap::RegistrationPointType::kVoxels;
    double layer_B_pose[4], layer_A_pose[4];
    for (int a = 0; a < 4; ++a) layer_B_pose[a] = truth[0][a], layer_A_pose[a] = truth[1][a] + drift[1][a];
    problem.AddParameterBlock(layer_B_pose, 4);
    problem.SetParameterBlockConstant(layer_B_pose);
    problem.AddParameterBlock(layer_A_pose, 4);
    VoxgraphSubmap::ConstPtr submap_B = submaps[0], submap_A = submaps[1];
#ifdef VGX_CALLERS_GPU
    ceres::CostFunction* cost_function = voxgraph_amd::MakeGpuRegistrationCostFunction(submap_B, submap_A, cost_config);
#else
    ceres::CostFunction* cost_function = new voxgraph::RegistrationCostFunction(submap_B, submap_A, cost_config);
#endif
    problem.AddResidualBlock(cost_function, nullptr, layer_B_pose, layer_A_pose);
    ceres::Solve(ceres_options, &problem, &summary);
    std::printf("ALIGN iterations=%d final_cost=%.9e pose %.9f %.9f %.9f %.9f\n", summary.num_iterations, summary.final_cost,
                layer_A_pose[0], layer_A_pose[1], layer_A_pose[2], layer_A_pose[3]);
    if (!(summary.final_cost < 0.2 * summary.initial_cost)) rc = 1;
  }
#ifdef VGX_CALLERS_GPU
  // GpuSubmapRegistry: a cached upload is believed only while its owner lives and its stamp is unchanged
  {
    voxgraph_amd::GpuSubmapRegistry& registry = voxgraph_amd::GpuSubmapRegistry::instance();
    const size_t size_before = registry.size();
    VoxgraphSubmap::ConstPtr kept = submaps[3];
    const vgx_submap first = registry.handleOf(kept);
    const vgx_submap again = registry.handleOf(kept);                 // same object, same stamp: the cached upload
    // the same object FINISHED AGAIN with another block in its layers: the stamp changes, the upload must be redone
    {
      voxblox::BlockIndex extra;
      extra[0] = 7; extra[1] = 7; extra[2] = 7;
      submaps[3]->getTsdfMapPtr()->getTsdfLayerPtr()->allocateBlockPtrByIndex(extra);
      std::ostringstream sink;
      std::streambuf* old = std::cout.rdbuf(sink.rdbuf());
      submaps[3]->finishSubmap();
      std::cout.rdbuf(old);
    }
    const long stale_before = registry.staleReplaced();
    const vgx_submap refreshed = registry.handleOf(kept);
    const long stale_after = registry.staleReplaced();
    // a submap that dies without release(): its entry is swept at the next call
    size_t size_with_temp = 0;
    {
      const double pose[4] = {9.0, 9.0, 0.0, 0.0};
      VoxgraphSubmap::ConstPtr temp = make_submap(99, pose, 0.1f, 16, 1, 0.3f, 1.0f);
      (void)registry.handleOf(temp);
      size_with_temp = registry.size();
    }
    (void)registry.handleOf(kept);
    std::printf("REGISTRY cached=%d stamp_change_reuploaded=%d stale_replaced=%ld size_before=%zu size_with_temp=%zu size_after_sweep=%zu\n",
                (int)(first == again), (int)(refreshed != nullptr), stale_after - stale_before, size_before, size_with_temp,
                registry.size());
    if (first != again || stale_after - stale_before != 1 || size_with_temp != size_before + 1 || registry.size() != size_before) rc = 1;
  }
  voxgraph_amd::GpuSubmapRegistry::instance().clear();
  vgx_ctx_destroy(ctx);
#endif
  return rc;
}
