// C++ end-to-end check of the host-side mirrors (voxgraph_amd/cpp/*.h) against closed
// forms, through the C ABI, the way the reference's C++ would call them.  Built and run
// by tests/test_cpp_adapter.py (g++ + libvoxgraph_amd.so; Ceres replaced by tests/stubs).
#include <cmath>
#include <cstdio>
#include <random>
#include <vector>

#include "gpu_fast_tsdf_integrator.h"
#include "gpu_registration_batch.h"
#include "gpu_registration_batch_multi.h"
#include "gpu_registration_cost_function.h"

static int fail(const char* what) {
  std::printf("FAIL: %s\n", what);
  return 1;
}

int main() {
  vgx_ctx ctx = nullptr;
  if (vgx_ctx_create(0, &ctx) != VGX_OK) return fail(vgx_last_error(nullptr));
  // ---- a planar ESDF: d(p) = n.p - c, trilinear interpolation is exact --------------
  const float vs = 0.1f;
  const int vps = 16, nb = 4 * 4 * 4;
  const float n[3] = {0.36f, 0.48f, 0.8f}, c = 0.2f;
  std::vector<int32_t> block_index;
  for (int x = -2; x < 2; ++x)
    for (int y = -2; y < 2; ++y)
      for (int z = -2; z < 2; ++z) { block_index.push_back(x); block_index.push_back(y); block_index.push_back(z); }
  const size_t nv = (size_t)nb * vps * vps * vps;
  std::vector<float> esdf(nv), tsdf(nv), w(nv, 10.0f);
  std::vector<uint8_t> obs(nv, 1);
  for (int b = 0; b < nb; ++b)
    for (int lin = 0; lin < vps * vps * vps; ++lin) {
      int v[3] = {lin % vps, (lin / vps) % vps, lin / (vps * vps)};
      float p[3];
      for (int a = 0; a < 3; ++a) p[a] = (float)block_index[3 * b + a] * (vps * vs) + ((float)v[a] + 0.5f) * vs;
      float d = n[0] * p[0] + n[1] * p[1] + n[2] * p[2] - c;
      esdf[(size_t)b * 4096 + lin] = d;
      tsdf[(size_t)b * 4096 + lin] = std::fmax(-0.3f, std::fmin(0.3f, d));
    }
  vgx_submap sm = nullptr;
  if (vgx_submap_create(ctx, 0, vs, vps, nb, block_index.data(), tsdf.data(), w.data(), esdf.data(),
                        obs.data(), &sm) != VGX_OK)
    return fail(vgx_last_error(ctx));
  std::mt19937 rng(1);
  std::uniform_real_distribution<float> U(-1.2f, 1.2f), Uw(1.5f, 10.0f), Ud(-0.3f, 0.3f);
  const int N = 5000;
  std::vector<float> xyz(3 * N), dist(N), wt(N);
  double sum_w = 0;
  for (int i = 0; i < N; ++i) {
    for (int a = 0; a < 3; ++a) xyz[3 * i + a] = U(rng);
    dist[i] = Ud(rng);
    wt[i] = Uw(rng);
    sum_w += wt[i];
  }
  if (vgx_submap_set_points(sm, VGX_POINTS_ISOSURFACE, N, xyz.data(), dist.data(), wt.data(),
                            VGX_POINTS_KEEP_ORDER) != VGX_OK)
    return fail(vgx_last_error(ctx));
  {
    voxgraph_amd::GpuRegistrationCostFunction::Config config;  // reference defaults
    voxgraph_amd::GpuRegistrationCostFunction cost(ctx, sm, sm, config);
    const ceres::CostFunction& base = cost;  // called through the Ceres interface
    if (base.num_residuals() != N || base.parameter_block_sizes().size() != 2 ||
        base.parameter_block_sizes()[0] != 4 || base.parameter_block_sizes()[1] != 4)
      return fail("sizes");
    double ref_pose[4] = {0.3, -0.2, 0.1, 0.4}, read_pose[4] = {0.1, 0.25, -0.15, -0.3};
    double* params[2] = {ref_pose, read_pose};
    std::vector<double> r(N), jo(4 * N), je(4 * N);
    double* jac[2] = {jo.data(), je.data()};
    if (!base.Evaluate(params, r.data(), jac)) return fail("Evaluate returned false");
    // closed form: p' = R(psi_e)^T (R(psi_o) p + t_o - t_e); r = (d - (n.p' - c)) w N/sum(w)
    const double F = N / sum_w, co = std::cos(ref_pose[3]), so = std::sin(ref_pose[3]);
    const double ce = std::cos(read_pose[3]), se = std::sin(read_pose[3]);
    double worst = 0;
    for (int i = 0; i < N; ++i) {
      double x = xyz[3 * i], y = xyz[3 * i + 1], z = xyz[3 * i + 2];
      double wx = co * x - so * y + ref_pose[0] - read_pose[0];
      double wy = so * x + co * y + ref_pose[1] - read_pose[1];
      double wz = z + ref_pose[2] - read_pose[2];
      double px = ce * wx + se * wy, py = -se * wx + ce * wy, pz = wz;
      double want = (dist[i] - (n[0] * px + n[1] * py + n[2] * pz - c)) * wt[i] * F;
      worst = std::fmax(worst, std::fabs(r[i] - want) / (wt[i] * F));
      // d r / d z_ref = -w F n_z ; d r / d z_read = +w F n_z
      if (std::fabs(jo[4 * i + 2] + wt[i] * F * n[2]) > 2e-3 * wt[i] * F) return fail("jac_ref z");
      if (std::fabs(je[4 * i + 2] - wt[i] * F * n[2]) > 2e-3 * wt[i] * F) return fail("jac_read z");
    }
    if (worst > 5e-5) return fail("residual closed form");
    // jacobians == nullptr and a null block (constant parameter block)
    std::vector<double> r2(N);
    if (!base.Evaluate(params, r2.data(), nullptr) || r2 != r) return fail("null jacobians");
    double* jac1[2] = {nullptr, je.data()};
    if (!base.Evaluate(params, r2.data(), jac1) || r2 != r) return fail("null block");
    std::printf("REG adapter ok: %d residuals, worst |dr|/(wF) = %.2e\n", N, worst);
    // ---- batched Ceres path: EvaluationCallback + 9-residual compressed block -----------
    voxgraph_amd::GpuRegistrationBatch batch(ctx);
    ceres::CostFunction* block = batch.AddConstraint(cost.handle(), ref_pose, read_pose);
    ceres::CostFunction* block2 = batch.AddConstraint(cost.handle(), read_pose, ref_pose);
    batch.Finalize();
    ceres::EvaluationCallback& cb = batch;
    cb.PrepareForEvaluation(true, true);
    double rc[9], jc0[36], jc1[36];
    double* jcs[2] = {jc0, jc1};
    if (block->num_residuals() != 9 || !block->Evaluate(params, rc, jcs)) return fail("compressed block");
    double cost_full = 0, cost_c = 0, g_full[8] = {0}, g_c[8] = {0};
    for (int i = 0; i < N; ++i) {
      cost_full += r[i] * r[i];
      for (int k2 = 0; k2 < 4; ++k2) {
        g_full[k2] += jo[4 * i + k2] * r[i];
        g_full[4 + k2] += je[4 * i + k2] * r[i];
      }
    }
    for (int m = 0; m < 9; ++m) {
      cost_c += rc[m] * rc[m];
      for (int k2 = 0; k2 < 4; ++k2) {
        g_c[k2] += jc0[4 * m + k2] * rc[m];
        g_c[4 + k2] += jc1[4 * m + k2] * rc[m];
      }
    }
    if (std::fabs(cost_c - cost_full) > 1e-6 * cost_full) return fail("compressed cost");
    for (int k2 = 0; k2 < 8; ++k2)
      if (std::fabs(g_c[k2] - g_full[k2]) > 1e-6 * (std::fabs(g_full[k2]) + 1e-3 * cost_full))
        return fail("compressed gradient");
    // without new_evaluation_point, at the same poses, the cache answers (no pass is run) ...
    double before = rc[8];
    const long passes = batch.full_evaluations() + batch.cost_only_evaluations();
    cb.PrepareForEvaluation(false, false);
    if (!block->Evaluate(params, rc, nullptr) || rc[8] != before || batch.full_evaluations() + batch.cost_only_evaluations() != passes)
      return fail("cache");
    // ... but never for another point: a pose moved WITHOUT an announcement (Problem::Evaluate and Covariance::Compute do
    // not call the callback of Solver::Options) is noticed by the block itself, which has the batch evaluated there
    read_pose[0] += 0.05;
    if (!block->Evaluate(params, rc, nullptr)) return fail("unannounced evaluation");
    double c2 = 0;
    for (int m = 0; m < 9; ++m) c2 += rc[m] * rc[m];
    if (std::fabs(c2 - cost_c) < 1e-9 * cost_c || batch.unannounced_evaluations() != 1) return fail("stale cache served");
    // the announced (cost-only) re-evaluation at that point gives the same cost
    cb.PrepareForEvaluation(false, true);
    double rc2[9], c3 = 0;
    if (!block->Evaluate(params, rc2, nullptr)) return fail("re-evaluation");
    for (int m = 0; m < 9; ++m) c3 += rc2[m] * rc2[m];
    if (std::fabs(c3 - c2) > 1e-6 * c2) return fail("re-evaluation: another cost");
    // parameters that are neither the cached point nor what the user's blocks hold: an evaluation failure, not a guess
    {
      double elsewhere[4] = {read_pose[0] + 1.0, read_pose[1], read_pose[2], read_pose[3]};
      const double* p2[2] = {params[0] == read_pose ? elsewhere : params[0], params[1] == read_pose ? elsewhere : params[1]};
      if (block->Evaluate(p2, rc2, nullptr)) return fail("a value for parameters nobody holds");
    }
    read_pose[0] -= 0.05;
    std::printf("batched Ceres path ok: cost %.6f == %.6f\n", cost_c, cost_full);
    // ---- the same two constraints sharded over two contexts (one device here; one per GPU
    // in production): identical 9-residual blocks, and the fused buffer of the in-process sum
    {
      vgx_ctx ctx_b = nullptr;
      if (vgx_ctx_create(0, &ctx_b) != VGX_OK) return fail("second context");
      vgx_submap sm_b = nullptr;
      if (vgx_submap_create(ctx_b, 0, vs, vps, nb, block_index.data(), tsdf.data(), w.data(), esdf.data(),
                            obs.data(), &sm_b) != VGX_OK ||
          vgx_submap_set_points(sm_b, VGX_POINTS_ISOSURFACE, N, xyz.data(), dist.data(), wt.data(),
                                VGX_POINTS_KEEP_ORDER) != VGX_OK)
        return fail(vgx_last_error(ctx_b));
      voxgraph_amd::GpuRegistrationCostFunction cost_b(ctx_b, sm_b, sm_b, config);
      std::vector<vgx_ctx> gpus = {ctx, ctx_b};
      voxgraph_amd::GpuRegistrationBatchMulti multi(gpus);
      std::vector<int32_t> plan = multi.PlanShards({N, N});
      if (plan.size() != 2 || plan[0] == plan[1]) return fail("LPT plan");
      ceres::CostFunction* m1 = multi.AddConstraint(plan[0] == 0 ? cost.handle() : cost_b.handle(), ref_pose, read_pose);
      ceres::CostFunction* m2 = multi.AddConstraint(plan[1] == 0 ? cost.handle() : cost_b.handle(), read_pose, ref_pose);
      multi.Finalize();
      ceres::EvaluationCallback& mcb = multi;
      mcb.PrepareForEvaluation(true, true);
      cb.PrepareForEvaluation(true, true);
      double r_s[9], r_m[9], j_s0[36], j_s1[36], j_m0[36], j_m1[36];
      double* js[2] = {j_s0, j_s1};
      double* jm[2] = {j_m0, j_m1};
      if (!block->Evaluate(params, r_s, js) || !m1->Evaluate(params, r_m, jm)) return fail("multi block");
      for (int q = 0; q < 9; ++q)
        if (r_s[q] != r_m[q]) return fail("multi block differs from the single-GPU block");
      for (int q = 0; q < 36; ++q)
        if (j_s0[q] != j_m0[q] || j_s1[q] != j_m1[q]) return fail("multi Jacobian differs");
      double* params_mirrored[2] = {read_pose, ref_pose};   // block2 / m2 were added with the blocks in this order
      if (!block2->Evaluate(params_mirrored, r_s, nullptr) || !m2->Evaluate(params_mirrored, r_m, nullptr)) return fail("multi block 2");
      for (int q = 0; q < 9; ++q)
        if (r_s[q] != r_m[q]) return fail("multi block 2 differs");
      double two_poses[8];
      for (int q = 0; q < 4; ++q) {
        two_poses[q] = ref_pose[q];
        two_poses[4 + q] = read_pose[q];
      }
      std::vector<double> fused = multi.EvaluateFused(two_poses, 2);
      double c_m = 0;
      for (int q = 0; q < 9; ++q) c_m += r_m[q] * r_m[q];
      // fused[0] = cost of both constraints; constraint 1's cost is cost_full (same residuals)
      if (fused.size() != 1 + 20 * 2 + 16 * 2 || std::fabs(fused[0] - (cost_full + c_m)) > 1e-6 * fused[0])
        return fail("multi fused cost");
      std::printf("multi-context Ceres path ok: fused cost %.6f over 2 contexts\n", fused[0]);
      delete m1;
      delete m2;
      // multi and cost_b go at scope exit; the submap and the second context outlive them
      // (process exit reclaims them: this is a smoke test)
    }
    delete block;
    delete block2;
  }
  // ---- TSDF adapter: one ray through an empty layer -----------------------------------
  {
    const int32_t box_min[3] = {-4, -4, -4}, box_dim[3] = {8, 8, 8};
    voxgraph_amd::GpuTsdfLayer layer(ctx, 0.1f, 16, box_min, box_dim, 64);
    auto cfg = voxgraph_amd::GpuFastTsdfIntegrator::defaultConfig();
    cfg.default_truncation_distance = 0.3f;
    cfg.use_const_weight = 1;
    cfg.use_weight_dropoff = 0;
    cfg.max_ray_length_m = 10.0f;
    voxgraph_amd::GpuFastTsdfIntegrator integ(ctx, cfg, &layer);
    integ.setLayer(&layer);
    const float T[7] = {1, 0, 0, 0, 0.05f, 0.05f, 0.05f};
    const float p[3] = {2.0f, 0.0f, 0.0f};
    integ.integratePointCloud(T, p, nullptr, 1);
    int32_t nblk = layer.getNumberOfAllocatedBlocks();
    std::vector<int32_t> bi(3 * nblk);
    std::vector<float> d((size_t)nblk * 4096), ww((size_t)nblk * 4096);
    if (vgx_tsdf_layer_download(layer.handle(), bi.data(), d.data(), ww.data(), nullptr) != VGX_OK)
      return fail("tsdf download");
    int updated = 0;
    for (int b = 0; b < nblk; ++b)
      for (int lin = 0; lin < 4096; ++lin)
        if (ww[(size_t)b * 4096 + lin] > 0) {
          int ix = bi[3 * b] * 16 + lin % 16;
          double sdf = 2.0 - ((ix + 0.5) * 0.1 - 0.05);
          double want = std::fmax(-0.3, std::fmin(0.3, sdf));
          if (std::fabs(d[(size_t)b * 4096 + lin] - want) > 1e-5) return fail("tsdf value");
          ++updated;
        }
    if (updated != 24 || nblk != 2) return fail("tsdf voxel count");
    std::printf("TSDF adapter ok: %d voxels in %d blocks\n", updated, nblk);
  }
  vgx_submap_destroy(sm);
  vgx_ctx_destroy(ctx);
  std::printf("ADAPTER_SMOKE_OK\n");
  return 0;
}
