// End-to-end in C++, the way voxgraph would run it (pose_graph.cpp:74-106): a ceres::Problem with the
// registration constraints of a two-submap graph, first pose constant (pose_graph_interface.cpp:30-32),
// ceres::Solve -- once per integration route of INTEGRATION.md:
//   1. drop-in: one voxgraph_amd::GpuRegistrationCostFunction per constraint (N residuals each, f64
//      residuals and Jacobians back over PCIe on every Evaluate),
//   2. batched: voxgraph_amd::GpuRegistrationBatch as Solver::Options::evaluation_callback, one fused
//      GPU pass per solver evaluation, each constraint a 9-residual block with the same normal equations,
//   3. multi-GPU: voxgraph_amd::GpuRegistrationBatchMulti over two contexts (one device here),
//   4. rows: voxgraph_amd::GpuRegistrationRows -- the reference's own N-residual blocks, all of them evaluated by one launch per
//      solver evaluation and fetched slice by slice (SURVEY 8b's vgx_reg_fetch): the drop-in route's VALUES, so its very path.
// The three routes minimise the same objective, so they must end in the same pose: within 1 mm and
// 0.01 degree of each other (north_star's end-pose tolerance), and at the true pose within a fraction
// of a voxel.  Ceres itself is absent from this image: tests/stubs/ceres/ceres.h supplies Problem /
// Solve with Ceres' conventions.  Built and run by tests/test_cpp_adapter.py.
#include <cmath>
#include <cstdio>
#include <memory>
#include <vector>

#include "gpu_registration_batch.h"
#include "gpu_registration_batch_multi.h"
#include "gpu_registration_cost_function.h"
#include "gpu_registration_rows.h"

namespace {
float scene_sdf(float x, float y, float z) {
  const float sphere = std::sqrt((x - 3.4f) * (x - 3.4f) + (y - 3.0f) * (y - 3.0f) + (z - 1.9f) * (z - 1.9f)) - 1.3f;
  const float ground = z - 0.45f;
  const float bx = std::fabs(x - 1.6f) - 0.6f, by = std::fabs(y - 4.3f) - 0.9f, bz = std::fabs(z - 1.0f) - 0.8f;
  const float ox = std::fmax(bx, 0.0f), oy = std::fmax(by, 0.0f), oz = std::fmax(bz, 0.0f);
  const float box = std::sqrt(ox * ox + oy * oy + oz * oz) + std::fmin(std::fmax(bx, std::fmax(by, bz)), 0.0f);
  return std::fmin(std::fmin(sphere, ground), box);
}

// a finished submap (TSDF + ESDF + kVoxels points) of the scene, sampled at mission pose {x,y,z,yaw}
vgx_submap make_submap(vgx_ctx ctx, int id, const double pose[4]) {
  const float vs = 0.1f;
  const int vps = 16, side = 4, nb = side * side * side;
  std::vector<int32_t> bi;
  for (int x = 0; x < side; ++x)
    for (int y = 0; y < side; ++y)
      for (int z = 0; z < side; ++z) {
        bi.push_back(x);
        bi.push_back(y);
        bi.push_back(z);
      }
  const size_t nv = (size_t)nb * 4096;
  std::vector<float> td(nv), tw(nv), ed(nv);
  std::vector<uint8_t> eo(nv);
  const float c = (float)std::cos(pose[3]), s = (float)std::sin(pose[3]);
  for (int b = 0; b < nb; ++b)
    for (int lin = 0; lin < 4096; ++lin) {
      const int v[3] = {lin % 16, (lin / 16) % 16, lin / 256};
      float p[3];
      for (int a = 0; a < 3; ++a) p[a] = (float)bi[3 * b + a] * (vps * vs) + ((float)v[a] + 0.5f) * vs;
      const float wx = c * p[0] - s * p[1] + (float)pose[0], wy = s * p[0] + c * p[1] + (float)pose[1];
      const float d = scene_sdf(wx, wy, p[2] + (float)pose[2]);
      const size_t at = (size_t)b * 4096 + lin;
      td[at] = std::fmax(-0.3f, std::fmin(0.3f, d));
      tw[at] = std::fabs(d) < 0.6f ? 10.0f : 0.0f;
      ed[at] = std::fmax(-2.0f, std::fmin(2.0f, d));
      eo[at] = std::fabs(d) <= 2.0f ? 1 : 0;
    }
  vgx_submap sm = nullptr;
  if (vgx_submap_create(ctx, id, vs, vps, nb, bi.data(), td.data(), tw.data(), ed.data(), eo.data(), &sm) != VGX_OK ||
      vgx_submap_extract_voxel_points(sm, 1.0, 0.3, 1, nullptr) != VGX_OK) {
    std::printf("FAIL: %s\n", vgx_last_error(ctx));
    return nullptr;
  }
  return sm;
}

void report(const char* route, const ceres::Solver::Summary& s, const double b[4]) {
  std::printf("%-10s %2d iterations, %2d evaluations, cost %.6e -> %.6e (%s): pose %.6f %.6f %.6f %.7f\n", route,
              s.num_iterations, s.num_evaluations, s.initial_cost, s.final_cost, s.termination, b[0], b[1], b[2], b[3]);
}
}  // namespace

int main() {
  vgx_ctx ctx = nullptr, ctx2 = nullptr;
  if (vgx_ctx_create(0, &ctx) != VGX_OK || vgx_ctx_create(0, &ctx2) != VGX_OK) {
    std::printf("FAIL: %s\n", vgx_last_error(nullptr));
    return 1;
  }
  const double a_true[4] = {0.0, 0.0, 0.0, 0.0}, b_true[4] = {0.9, 0.6, 0.1, 0.12};
  const double b_start[4] = {0.9 + 0.15, 0.6 - 0.10, 0.1 + 0.05, 0.12 + 0.03};
  vgx_submap A = make_submap(ctx, 0, a_true), B = make_submap(ctx, 1, b_true);
  vgx_submap A2 = make_submap(ctx2, 0, a_true), B2 = make_submap(ctx2, 1, b_true);
  if (!A || !B || !A2 || !B2) return 1;
  voxgraph_amd::GpuRegistrationCostFunction::Config cfg;
  cfg.registration_point_type = VGX_POINTS_VOXELS;
  ceres::Solver::Options options;                       // pose_graph.cpp:90-97 (tolerances: Ceres defaults)
  options.max_num_iterations = 50;
  double end[4][4];
  int dropin_iterations = 0;
  double dropin_final_cost = 0;
  // what Ceres evaluates WITHOUT announcing it to an evaluation callback (the callback sits in Solver::Options): a
  // residual-only Problem::Evaluate (PoseGraph::getVisualizationEdges, pose_graph.cpp:173-174) and Covariance::Compute
  // (getEdgeCovarianceMap, pose_graph.cpp:140) -- at the final point, and at a point no solver evaluation saw
  double final_cost_again[4] = {0, 0, 0, 0}, probe_cost[4] = {0, 0, 0, 0}, covariance[4][16];
  const double probe_offset[4] = {0.02, -0.015, 0.01, 0.004};
  auto unannounced = [&](int route, ceres::Problem* problem, double* b) -> bool {
    bool ok = problem->Evaluate(ceres::Problem::EvaluateOptions(), &final_cost_again[route], nullptr, nullptr, nullptr);
    double keep[4];
    for (int k = 0; k < 4; ++k) keep[k] = b[k], b[k] = end[0][k] + probe_offset[k];   // the same point on every route
    ok = ok && problem->Evaluate(ceres::Problem::EvaluateOptions(), &probe_cost[route], nullptr, nullptr, nullptr);
    for (int k = 0; k < 4; ++k) b[k] = end[0][k];
    ceres::Covariance::Options covariance_options;
    ceres::Covariance cov(covariance_options);
    std::vector<std::pair<const double*, const double*> > blocks;
    blocks.emplace_back(b, b);
    ok = ok && cov.Compute(blocks, problem) && cov.GetCovarianceBlock(b, b, covariance[route]);
    for (int k = 0; k < 4; ++k) b[k] = keep[k];
    return ok;
  };
  // ---- 1. drop-in cost functions -----------------------------------------------------------------------
  {
    double a[4], b[4];
    for (int k = 0; k < 4; ++k) a[k] = a_true[k], b[k] = b_start[k];
    ceres::Problem problem;
    problem.AddResidualBlock(new voxgraph_amd::GpuRegistrationCostFunction(ctx, A, B, cfg), nullptr, a, b);
    problem.AddResidualBlock(new voxgraph_amd::GpuRegistrationCostFunction(ctx, B, A, cfg), nullptr, b, a);
    problem.SetParameterBlockConstant(a);
    ceres::Solver::Summary summary;
    ceres::Solve(options, &problem, &summary);
    report("drop-in", summary, b);
    for (int k = 0; k < 4; ++k) end[0][k] = b[k];
    dropin_iterations = summary.num_iterations;
    dropin_final_cost = summary.final_cost;
    if (!(summary.final_cost < 0.05 * summary.initial_cost)) return std::printf("FAIL: drop-in did not converge\n"), 1;
    if (!unannounced(0, &problem, b)) return std::printf("FAIL: drop-in: Problem::Evaluate / Covariance failed\n"), 1;
    if (std::fabs(final_cost_again[0] - summary.final_cost) > 1e-9 * summary.final_cost) return std::printf("FAIL: drop-in: cost at the final point\n"), 1;
  }
  // ---- 2. batched evaluation callback ----------------------------------------------------------------------
  {
    double a[4], b[4];
    for (int k = 0; k < 4; ++k) a[k] = a_true[k], b[k] = b_start[k];
    voxgraph_amd::GpuRegistrationCostFunction ab(ctx, A, B, cfg), ba(ctx, B, A, cfg);   // own the vgx_reg handles
    voxgraph_amd::GpuRegistrationBatch batch(ctx);
    ceres::Problem problem;
    problem.AddResidualBlock(batch.AddConstraint(ab.handle(), a, b), nullptr, a, b);
    problem.AddResidualBlock(batch.AddConstraint(ba.handle(), b, a), nullptr, b, a);
    batch.Finalize();
    problem.SetParameterBlockConstant(a);
    ceres::Solver::Options o2 = options;
    o2.evaluation_callback = &batch;
    ceres::Solver::Summary summary;
    ceres::Solve(o2, &problem, &summary);
    report("batched", summary, b);
    for (int k = 0; k < 4; ++k) end[1][k] = b[k];
    // every trial step was a cost-only evaluation (registration_cost_function.cpp:179: no Jacobian work when Ceres passes
    // none) and took the cost-only route; every accepted step's Jacobians one full pass at the same point
    std::printf("batched: %d cost-only evaluations by the solver, %ld through vgx_reg_batch_evaluate_cost; %d with Jacobians, %ld "
                "full passes\n", summary.num_cost_only_evaluations, batch.cost_only_evaluations(),
                summary.num_jacobian_evaluations, batch.full_evaluations());
    if (summary.num_cost_only_evaluations < 1 || batch.cost_only_evaluations() != summary.num_cost_only_evaluations ||
        batch.full_evaluations() != summary.num_jacobian_evaluations)
      return std::printf("FAIL: cost-only evaluations did not take the cost-only route\n"), 1;
    // Unannounced evaluations: the blocks notice that their parameters are not the cached point's (or that Jacobians are
    // asked of a cost-only cache) and have the batch evaluated where the user's parameter blocks are
    if (!unannounced(1, &problem, b)) return std::printf("FAIL: batched: Problem::Evaluate / Covariance failed\n"), 1;
    std::printf("batched: %ld evaluations the blocks asked for themselves (Problem::Evaluate x 2, Covariance::Compute)\n",
                batch.unannounced_evaluations());
    if (std::fabs(final_cost_again[1] - summary.final_cost) > 1e-9 * summary.final_cost)
      return std::printf("FAIL: batched: Problem::Evaluate after the solve is not the final cost (%.9e vs %.9e): a stale cache\n",
                         final_cost_again[1], summary.final_cost), 1;
    if (batch.unannounced_evaluations() < 2) return std::printf("FAIL: batched: no unannounced evaluation was noticed\n"), 1;
  }
  // ---- 3. two contexts ----------------------------------------------------------------------------------------
  {
    double a[4], b[4];
    for (int k = 0; k < 4; ++k) a[k] = a_true[k], b[k] = b_start[k];
    voxgraph_amd::GpuRegistrationCostFunction ab(ctx, A, B, cfg), ba(ctx2, B2, A2, cfg);
    std::vector<vgx_ctx> gpus = {ctx, ctx2};
    voxgraph_amd::GpuRegistrationBatchMulti batch(gpus);
    ceres::Problem problem;
    problem.AddResidualBlock(batch.AddConstraint(ab.handle(), a, b), nullptr, a, b);
    problem.AddResidualBlock(batch.AddConstraint(ba.handle(), b, a), nullptr, b, a);
    batch.Finalize();
    problem.SetParameterBlockConstant(a);
    ceres::Solver::Options o3 = options;
    o3.evaluation_callback = &batch;
    ceres::Solver::Summary summary;
    ceres::Solve(o3, &problem, &summary);
    report("multi", summary, b);
    for (int k = 0; k < 4; ++k) end[2][k] = b[k];
    if (batch.cost_only_evaluations() != summary.num_cost_only_evaluations || batch.full_evaluations() != summary.num_jacobian_evaluations)
      return std::printf("FAIL: multi: cost-only evaluations did not take the cost-only route\n"), 1;
    if (!unannounced(2, &problem, b)) return std::printf("FAIL: multi: Problem::Evaluate / Covariance failed\n"), 1;
    if (std::fabs(final_cost_again[2] - summary.final_cost) > 1e-9 * summary.final_cost) return std::printf("FAIL: multi: stale cache\n"), 1;
  }
  // ---- 4. the reference's own blocks, one launch per evaluation ---------------------------------------------
  {
    double a[4], b[4];
    for (int k = 0; k < 4; ++k) a[k] = a_true[k], b[k] = b_start[k];
    voxgraph_amd::GpuRegistrationCostFunction ab(ctx, A, B, cfg), ba(ctx, B, A, cfg);
    voxgraph_amd::GpuRegistrationRows rows(ctx);
    ceres::Problem problem;
    ceres::CostFunction* f_ab = rows.AddConstraint(ab.handle(), a, b);
    ceres::CostFunction* f_ba = rows.AddConstraint(ba.handle(), b, a);
    if (f_ab->num_residuals() != ab.num_residuals() || f_ba->num_residuals() != ba.num_residuals())
      return std::printf("FAIL: rows: block sizes\n"), 1;
    problem.AddResidualBlock(f_ab, nullptr, a, b);
    problem.AddResidualBlock(f_ba, nullptr, b, a);
    rows.Finalize();
    problem.SetParameterBlockConstant(a);
    ceres::Solver::Options o4 = options;
    o4.evaluation_callback = &rows;
    ceres::Solver::Summary summary;
    ceres::Solve(o4, &problem, &summary);
    report("rows", summary, b);
    for (int k = 0; k < 4; ++k) end[3][k] = b[k];
    std::printf("rows: %d solver evaluations, %ld launches of the batch\n", summary.num_evaluations, rows.evaluations());
    // the drop-in route's values => its iterations, its cost, its end pose, to the last bit
    if (summary.num_iterations != dropin_iterations || summary.final_cost != dropin_final_cost)
      return std::printf("FAIL: rows: not the drop-in route's path (%d iterations, cost %.17g against %d, %.17g)\n", summary.num_iterations,
                         summary.final_cost, dropin_iterations, dropin_final_cost), 1;
    for (int k = 0; k < 4; ++k)
      if (end[3][k] != end[0][k]) return std::printf("FAIL: rows: end pose differs from the drop-in route's\n"), 1;
    if (rows.evaluations() > summary.num_evaluations) return std::printf("FAIL: rows: more launches than evaluations\n"), 1;
    if (!unannounced(3, &problem, b)) return std::printf("FAIL: rows: Problem::Evaluate / Covariance failed\n"), 1;
    if (final_cost_again[3] != summary.final_cost || probe_cost[3] != probe_cost[0])
      return std::printf("FAIL: rows: unannounced evaluations are not the drop-in route's values\n"), 1;
    for (int k = 0; k < 16; ++k)
      if (covariance[3][k] != covariance[0][k]) return std::printf("FAIL: rows: covariance differs from the drop-in route's\n"), 1;
  }
  // ---- the unannounced evaluations agree across the routes (the compressed blocks have the originals' normal equations) ----
  {
    double worst_cost = 0, worst_cov = 0, cov_scale = 0;
    for (int k = 0; k < 16; ++k) cov_scale = std::fmax(cov_scale, std::fabs(covariance[0][k]));
    for (int r = 1; r < 3; ++r) {
      worst_cost = std::fmax(worst_cost, std::fabs(probe_cost[r] - probe_cost[0]) / probe_cost[0]);
      for (int k = 0; k < 16; ++k) worst_cov = std::fmax(worst_cov, std::fabs(covariance[r][k] - covariance[0][k]) / cov_scale);
    }
    std::printf("unannounced evaluations: cost at a point no solve saw %.9e (routes agree to %.1e), covariance of the free pose "
                "(routes agree to %.1e of its largest entry %.3e)\n", probe_cost[0], worst_cost, worst_cov, cov_scale);
    if (!(probe_cost[0] > 0) || worst_cost > 1e-5 || !(cov_scale > 0) || worst_cov > 1e-4) return std::printf("FAIL: unannounced evaluations disagree\n"), 1;
  }
  // ---- same end pose on every route: 1 mm / 0.01 degree -------------------------------------------------------
  double worst_xyz = 0, worst_yaw = 0, from_truth = 0;
  for (int r = 1; r < 3; ++r)
    for (int k = 0; k < 4; ++k) {
      const double d = std::fabs(end[r][k] - end[0][k]);
      if (k < 3) worst_xyz = std::fmax(worst_xyz, d); else worst_yaw = std::fmax(worst_yaw, d);
    }
  for (int k = 0; k < 3; ++k) from_truth = std::fmax(from_truth, std::fabs(end[0][k] - b_true[k]));
  std::printf("routes agree to %.2e m / %.2e rad; end pose %.4f m from the truth, yaw error %.5f rad\n", worst_xyz,
              worst_yaw, from_truth, std::fabs(end[0][3] - b_true[3]));
  if (worst_xyz > 1e-3 || worst_yaw > 1.745e-4) return std::printf("FAIL: routes disagree\n"), 1;
  if (from_truth > 0.03 || std::fabs(end[0][3] - b_true[3]) > 0.005) return std::printf("FAIL: far from the truth\n"), 1;
  vgx_submap_destroy(A);
  vgx_submap_destroy(B);
  vgx_submap_destroy(A2);
  vgx_submap_destroy(B2);
  vgx_ctx_destroy(ctx2);
  vgx_ctx_destroy(ctx);
  std::printf("SOLVE_SMOKE_OK\n");
  return 0;
}
