// In-process drop-in check (TEST INFRASTRUCTURE; built into oracle/_ref/dropin_check where
// /root/reference exists, run on the GPU box by tests/test_dropin_cpp_gpu.py).
//
// One process holds BOTH implementations behind the same ceres::CostFunction interface:
//   * the reference's voxgraph::RegistrationCostFunction, compiled from /root/reference, reading the
//     reference's own VoxgraphSubmap objects (finishSubmap() has produced the registration points);
//   * voxgraph_amd::GpuRegistrationCostFunction over libvoxgraph_amd.so, fed through
//     voxgraph_amd/cpp/voxgraph_submap_bridge.h from those same VoxgraphSubmap objects --
// exactly the swap INTEGRATION.md describes for registration_constraint.cpp:33-35 -- and every
// residual and Jacobian entry of Evaluate() is compared.
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <iostream>
#include <memory>
#include <random>
#include <sstream>
#include <vector>

#include <cblox/core/tsdf_esdf_submap.h>
#include <ceres/ceres.h>

#include "voxgraph/backend/constraint/cost_functions/registration_cost_function.h"
#include "voxgraph/frontend/submap_collection/voxgraph_submap.h"

#include "gpu_registration_cost_function.h"
#include "voxgraph_submap_bridge.h"

using voxgraph::RegistrationCostFunction;
using voxgraph::VoxgraphSubmap;

namespace {
// analytic scene in the mission frame: sphere + ground + a box, sampled into a TSDF/ESDF pair
float scene_sdf(float x, float y, float z) {
  const float sphere = std::sqrt((x - 1.7f) * (x - 1.7f) + (y - 1.5f) * (y - 1.5f) + (z - 1.2f) * (z - 1.2f)) - 0.9f;
  const float ground = z - 0.35f;
  const float bx = std::fabs(x - 0.9f) - 0.35f, by = std::fabs(y - 2.4f) - 0.5f, bz = std::fabs(z - 0.6f) - 0.4f;
  const float ox = std::fmax(bx, 0.0f), oy = std::fmax(by, 0.0f), oz = std::fmax(bz, 0.0f);
  const float box = std::sqrt(ox * ox + oy * oy + oz * oz) + std::fmin(std::fmax(bx, std::fmax(by, bz)), 0.0f);
  return std::fmin(std::fmin(sphere, ground), box);
}

std::shared_ptr<VoxgraphSubmap> make_submap(unsigned id, const double pose[4], float voxel_size, int vps,
                                            int blocks_per_side, float trunc, float esdf_max) {
  voxblox::Transformation::Vector6 v;
  v[0] = pose[0];
  v[1] = pose[1];
  v[2] = pose[2];
  v[3] = 0;
  v[4] = 0;
  v[5] = pose[3];
  const voxblox::Transformation T_M_S = voxblox::Transformation::exp(v);
  voxblox::Layer<voxblox::TsdfVoxel> tsdf(voxel_size, vps);
  for (int bx = 0; bx < blocks_per_side; ++bx)
    for (int by = 0; by < blocks_per_side; ++by)
      for (int bz = 0; bz < blocks_per_side; ++bz) {
        voxblox::BlockIndex idx;
        idx[0] = bx;
        idx[1] = by;
        idx[2] = bz;
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

struct Worst {
  double abs = 0, rel = 0;
  long long differing = 0, n = 0;
  void add(double got, double want, double scale) {
    ++n;
    if (got != want) ++differing;
    const double e = std::fabs(got - want);
    if (e > abs) abs = e;
    const double r = e / std::fmax(std::fabs(want), 1e-3 * scale);
    if (r > rel) rel = r;
  }
};
}  // namespace

int main() {
  vgx_ctx ctx = nullptr;
  if (vgx_ctx_create(0, &ctx) != VGX_OK) {
    std::fprintf(stderr, "no device\n");
    return 2;
  }
  const double pose_a[4] = {0.2, -0.1, 0.05, 0.1}, pose_b[4] = {0.9, 0.5, 0.0, -0.25};
  auto A = make_submap(7, pose_a, 0.1f, 16, 2, 0.3f, 1.0f);
  auto B = make_submap(9, pose_b, 0.1f, 16, 2, 0.3f, 1.0f);
  vgx_submap gA = voxgraph_amd::UploadFinishedSubmap(ctx, *A);
  vgx_submap gB = voxgraph_amd::UploadFinishedSubmap(ctx, *B);

  std::mt19937 rng(3);
  std::normal_distribution<double> noise(0.0, 1.0);
  int failures = 0, cases = 0;
  Worst worst;
  for (int point_type = 0; point_type < 2; ++point_type)
    for (int use_esdf = 0; use_esdf < 2; ++use_esdf)
      for (int sampled = 0; sampled < 3; ++sampled) {
        RegistrationCostFunction::Config rc;
        rc.registration_point_type = static_cast<VoxgraphSubmap::RegistrationPointType>(point_type);
        rc.use_esdf_distance = use_esdf != 0;
        rc.sampling_ratio = sampled == 0 ? -1.0f : (sampled == 1 ? 0.2f : 1.5f);  // all / down- / up-sampling
        rc.no_correspondence_cost = use_esdf ? 0.0 : 0.4;
        voxgraph_amd::GpuRegistrationCostFunction::Config gc;
        gc.registration_point_type = point_type;
        gc.use_esdf_distance = rc.use_esdf_distance;
        gc.sampling_ratio = rc.sampling_ratio;
        gc.no_correspondence_cost = rc.no_correspondence_cost;
        // the two objects a maintainer would choose between at registration_constraint.cpp:33-35
        std::unique_ptr<ceres::CostFunction> reference(new RegistrationCostFunction(A, B, rc));
        std::unique_ptr<ceres::CostFunction> gpu(new voxgraph_amd::GpuRegistrationCostFunction(ctx, gA, gB, gc));
        if (reference->num_residuals() != gpu->num_residuals() ||
            reference->parameter_block_sizes() != gpu->parameter_block_sizes()) {
          std::printf("FAIL sizes: %d vs %d residuals\n", reference->num_residuals(), gpu->num_residuals());
          ++failures;
          continue;
        }
        const int n = reference->num_residuals();
        for (int trial = 0; trial < 3; ++trial, ++cases) {   // successive calls continue the sampler streams
          double pa[4], pb[4];
          for (int k = 0; k < 4; ++k) {
            pa[k] = pose_a[k] + (k < 3 ? 0.05 : 0.03) * noise(rng);
            pb[k] = pose_b[k] + (k < 3 ? 0.05 : 0.03) * noise(rng);
          }
          const double* params[2] = {pa, pb};
          std::vector<double> r0(n), r1(n), j0a(4 * n), j0b(4 * n), j1a(4 * n), j1b(4 * n);
          double* jr[2] = {j0a.data(), j0b.data()};
          double* jg[2] = {j1a.data(), j1b.data()};
          if (trial == 2) jr[0] = jg[0] = nullptr;             // constant first block (pose_graph_interface.cpp:30-32)
          const bool ok0 = reference->Evaluate(params, r0.data(), jr);
          const bool ok1 = gpu->Evaluate(params, r1.data(), jg);
          if (ok0 != ok1) {
            std::printf("FAIL return value\n");
            ++failures;
            continue;
          }
          double scale_r = 0, scale_j = 0;
          for (int i = 0; i < n; ++i) scale_r = std::fmax(scale_r, std::fabs(r0[i]));
          for (int i = 0; i < 4 * n; ++i) scale_j = std::fmax(scale_j, std::fabs(j0b[i]));
          Worst w;
          for (int i = 0; i < n; ++i) w.add(r1[i], r0[i], scale_r);
          for (int i = 0; i < 4 * n; ++i) {
            if (jr[0]) w.add(j1a[i], j0a[i], scale_j);
            w.add(j1b[i], j0b[i], scale_j);
          }
          if (w.rel > 1e-4) ++failures;                        // north_star tolerance
          worst.abs = std::fmax(worst.abs, w.abs);
          worst.rel = std::fmax(worst.rel, w.rel);
          worst.differing += w.differing;
          worst.n += w.n;
        }
      }
  std::printf("DROPIN cases=%d failures=%d values=%lld differing=%lld worst_abs=%.3e worst_rel=%.3e\n", cases,
              failures, worst.n, worst.differing, worst.abs, worst.rel);
  vgx_submap_destroy(gA);
  vgx_submap_destroy(gB);
  vgx_ctx_destroy(ctx);
  return failures == 0 ? 0 : 1;
}
