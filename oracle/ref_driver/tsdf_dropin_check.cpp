// In-process TSDF drop-in check (TEST INFRASTRUCTURE; built into oracle/_build/tsdf_dropin_check by
// `make -C oracle`, run on the GPU box by tests/test_tsdf_dropin_gpu.py).
//
// voxgraph drives voxblox::FastTsdfIntegrator like this
// (voxgraph/src/frontend/measurement_processors/pointcloud_integrator.cpp:66-83):
//     integrator.reset(new FastTsdfIntegrator(config, layer)); integrator->setLayer(layer);
//     integrator->integratePointCloud(T_submap_sensor, pointcloud, colors);
// and finishSubmap() then reads the voxblox layer on the host (voxgraph_submap.cpp:84-107).
// This program does the same with voxgraph_amd::GpuFastTsdfIntegrator on an UNBOUNDED GpuTsdfLayer,
// hands the result to a voxblox::Layer<TsdfVoxel> (oracle/ref_shims) through
// voxgraph_amd/cpp/gpu_tsdf_layer_bridge.h, and compares it voxel for voxel -- distance, weight and
// colour bits, and the set of allocated blocks -- with the CPU restatement of voxblox's integrator
// (oracle/tsdf_oracle.c; voxblox itself is not vendored: PARITY UNPINNED) on the scans where the
// algorithm is order-independent: single-ray scans and scans of rays that share no voxel.
// The sensor wanders 60 m, so the layer has to grow far beyond any initial box.
#include <cmath>
#include <cstdio>
#include <cstring>
#include <map>
#include <random>
#include <vector>

#include <voxblox/core/layer.h>

#include "gpu_fast_tsdf_integrator.h"
#include "gpu_tsdf_layer_bridge.h"
#include "tsdf_oracle.h"

namespace {
struct Key {
  int x, y, z;
  bool operator<(const Key& o) const { return x != o.x ? x < o.x : (y != o.y ? y < o.y : z < o.z); }
};

struct Totals {
  long long scans = 0, voxels = 0, differing = 0, blocks_gpu = 0, blocks_cpu = 0;
};

// the oracle's layer against the voxblox layer the GPU contents were downloaded into
void compare(const orc_tsdf_layer* ol, const voxblox::Layer<voxblox::TsdfVoxel>& vl, int vps, Totals* t,
             const char* what) {
  const int n = orc_tsdf_layer_num_blocks(ol);
  const size_t vpb = (size_t)vps * vps * vps;
  std::vector<int32_t> bi(3 * (size_t)n);
  std::vector<float> d(vpb * n), w(vpb * n);
  std::vector<uint8_t> c(4 * vpb * n);
  orc_tsdf_layer_download(ol, bi.data(), d.data(), w.data(), c.data());
  voxblox::BlockIndexList gpu_blocks;
  vl.getAllAllocatedBlocks(&gpu_blocks);
  t->blocks_cpu = n;
  t->blocks_gpu = (long long)gpu_blocks.size();
  long long bad = 0;
  if ((size_t)n != gpu_blocks.size()) bad += std::llabs((long long)n - (long long)gpu_blocks.size());
  for (int b = 0; b < n; ++b) {
    voxblox::BlockIndex idx;
    idx[0] = bi[3 * b];
    idx[1] = bi[3 * b + 1];
    idx[2] = bi[3 * b + 2];
    if (!vl.hasBlock(idx)) {
      bad += (long long)vpb;
      continue;
    }
    const auto& block = vl.getBlockByIndex(idx);
    for (size_t lin = 0; lin < vpb; ++lin) {
      const voxblox::TsdfVoxel& v = block.getVoxelByLinearIndex(lin);
      const size_t at = (size_t)b * vpb + lin;
      const uint8_t col[4] = {v.color.r, v.color.g, v.color.b, v.color.a};
      if (std::memcmp(&v.distance, &d[at], 4) != 0 || std::memcmp(&v.weight, &w[at], 4) != 0 ||
          std::memcmp(col, &c[4 * at], 4) != 0)
        ++bad;
      ++t->voxels;
    }
  }
  if (bad) std::printf("  %s: %lld differing voxels (cpu blocks %d, gpu blocks %zu)\n", what, bad, n, gpu_blocks.size());
  t->differing += bad;
}

void quat_yaw_pitch(float yaw, float pitch, float q[4]) {
  // q = qz(yaw) * qy(pitch)
  const float cy = std::cos(yaw / 2), sy = std::sin(yaw / 2), cp = std::cos(pitch / 2), sp = std::sin(pitch / 2);
  q[0] = cy * cp;
  q[1] = -sy * sp;
  q[2] = cy * sp;
  q[3] = sy * cp;
}
}  // namespace

int main() {
  vgx_ctx ctx = nullptr;
  if (vgx_ctx_create(0, &ctx) != VGX_OK) {
    std::printf("FAIL: %s\n", vgx_last_error(nullptr));
    return 1;
  }
  Totals tot;
  const float vs = 0.2f;
  const int vps = 16;
  std::mt19937 rng(7);
  std::uniform_real_distribution<float> U(-1.0f, 1.0f), U01(0.0f, 1.0f);
  std::uniform_int_distribution<int> C(0, 255);
  long long growths = 0;
  // ---- A: the shipped integrator settings (voxgraph_mapper.yaml:21-28), one ray per scan ------------
  {
    voxgraph_amd::GpuFastTsdfIntegrator::Config gc = voxgraph_amd::GpuFastTsdfIntegrator::defaultConfig();
    orc_tsdf_config oc;
    orc_tsdf_config_default(&oc);
    gc.default_truncation_distance = oc.default_truncation_distance = 0.6f;
    gc.max_ray_length_m = oc.max_ray_length_m = 16.0f;
    gc.use_const_weight = oc.use_const_weight = 1;
    gc.use_weight_dropoff = oc.use_weight_dropoff = 1;
    gc.use_sparsity_compensation_factor = oc.use_sparsity_compensation_factor = 1;
    gc.sparsity_compensation_factor = oc.sparsity_compensation_factor = 20.0f;
    voxgraph_amd::GpuTsdfLayer gpu_layer(ctx, vs, vps);                 // no box, no pool size: unbounded
    voxgraph_amd::GpuFastTsdfIntegrator gpu(ctx, gc, &gpu_layer);
    gpu.setLayer(&gpu_layer);                                            // pointcloud_integrator.cpp:77
    orc_tsdf_layer* ol = orc_tsdf_layer_create(vs, vps);
    orc_tsdf_integrator* oi = orc_tsdf_integrator_create(&oc, ol);
    voxblox::Layer<voxblox::TsdfVoxel> host_layer(vs, vps);
    for (int k = 0; k < 240; ++k) {
      // the sensor walks 60 m along x (and back up in z): far outside any box a caller might have guessed
      float T[7];
      quat_yaw_pitch(0.4f * U(rng) + 0.02f * k, 0.2f * U(rng), T);
      T[4] = 0.25f * k + 0.3f * U(rng);
      T[5] = 3.0f * std::sin(0.05f * k);
      T[6] = 1.0f + 0.01f * k;
      float dir[3] = {U(rng), U(rng), 0.4f * U(rng)};
      const float nrm = std::sqrt(dir[0] * dir[0] + dir[1] * dir[1] + dir[2] * dir[2]) + 1e-6f;
      // lengths: mostly inside the ray limit, some beyond it (clearing rays), a few below the minimum
      const float len = k % 9 == 0 ? 16.0f + 8.0f * U01(rng) : (k % 31 == 0 ? 0.05f : 0.5f + 15.0f * U01(rng));
      const float p[3] = {dir[0] / nrm * len, dir[1] / nrm * len, dir[2] / nrm * len};
      const uint8_t col[4] = {(uint8_t)C(rng), (uint8_t)C(rng), (uint8_t)C(rng), 255};
      gpu.integratePointCloud(T, p, col, 1, /*freespace_points=*/k % 17 == 0);
      orc_tsdf_integrate(oi, T, p, col, 1, k % 17 == 0);
      ++tot.scans;
      if (k % 60 == 59) {
        voxgraph_amd::DownloadTsdfLayer(gpu_layer, &host_layer);
        compare(ol, host_layer, vps, &tot, "single-ray scans");
      }
    }
    growths += vgx_tsdf_layer_growths(gpu_layer.handle());
    // ---- C: hand the voxblox layer back to a FRESH GPU layer, continue integrating on it -------------
    voxgraph_amd::GpuTsdfLayer second(ctx, vs, vps);
    voxgraph_amd::UploadTsdfLayer(host_layer, &second);
    voxblox::Layer<voxblox::TsdfVoxel> round_trip(vs, vps);
    voxgraph_amd::DownloadTsdfLayer(second, &round_trip);
    compare(ol, round_trip, vps, &tot, "upload -> download round trip");
    gpu.setLayer(&second);
    for (int k = 0; k < 40; ++k) {
      float T[7];
      quat_yaw_pitch(U(rng), 0.1f * U(rng), T);
      T[4] = 60.0f + 0.5f * k;
      T[5] = -2.0f + 0.1f * k;
      T[6] = 1.5f;
      const float len = 2.0f + 12.0f * U01(rng);
      const float a = 3.14159f * U(rng);
      const float p[3] = {std::cos(a) * len, std::sin(a) * len, 0.3f * U(rng) * len};
      const uint8_t col[4] = {(uint8_t)C(rng), (uint8_t)C(rng), (uint8_t)C(rng), 255};
      gpu.integratePointCloud(T, p, col, 1);
      orc_tsdf_integrate(oi, T, p, col, 1, 0);
      ++tot.scans;
    }
    voxblox::Layer<voxblox::TsdfVoxel> after(vs, vps);
    voxgraph_amd::DownloadTsdfLayer(second, &after);
    compare(ol, after, vps, &tot, "scans continued on the uploaded layer");
    growths += vgx_tsdf_layer_growths(second.handle());
    orc_tsdf_integrator_destroy(oi);
    orc_tsdf_layer_destroy(ol);
  }
  // ---- B: many rays per scan that share no voxel (no carving: only the truncation band is touched) -----
  {
    voxgraph_amd::GpuFastTsdfIntegrator::Config gc = voxgraph_amd::GpuFastTsdfIntegrator::defaultConfig();
    orc_tsdf_config oc;
    orc_tsdf_config_default(&oc);
    gc.default_truncation_distance = oc.default_truncation_distance = 0.3f;
    gc.max_ray_length_m = oc.max_ray_length_m = 30.0f;
    gc.voxel_carving_enabled = oc.voxel_carving_enabled = 0;
    gc.use_const_weight = oc.use_const_weight = 0;
    gc.use_weight_dropoff = oc.use_weight_dropoff = 1;
    const float vs2 = 0.1f;
    voxgraph_amd::GpuTsdfLayer gpu_layer(ctx, vs2, vps);
    voxgraph_amd::GpuFastTsdfIntegrator gpu(ctx, gc, &gpu_layer);
    orc_tsdf_layer* ol = orc_tsdf_layer_create(vs2, vps);
    orc_tsdf_integrator* oi = orc_tsdf_integrator_create(&oc, ol);
    for (int scan = 0; scan < 6; ++scan) {
      // a 40 x 30 fan of rays whose end points are >= 1.5 m apart: bands of +-0.3 m never meet
      std::vector<float> pts;
      std::vector<uint8_t> cols;
      for (int i = 0; i < 40; ++i)
        for (int j = 0; j < 30; ++j) {
          const float y = -30.0f + 1.5f * i + 0.2f * U01(rng), z = -20.0f + 1.5f * j + 0.2f * U01(rng);
          const float x = 12.0f + 2.0f * scan + 0.3f * U01(rng);
          pts.push_back(x);
          pts.push_back(y);
          pts.push_back(z);
          for (int q = 0; q < 4; ++q) cols.push_back((uint8_t)C(rng));
        }
      const float T[7] = {1, 0, 0, 0, 0.05f + 3.0f * scan, 0.02f, 0.03f};
      // the reference's own call, with voxblox's types (pointcloud_integrator.cpp:83): Transformation, Pointcloud, Colors
      {
        voxblox::Transformation T_G_C(voxblox::Transformation::Rotation(T[0], T[1], T[2], T[3]),
                                      voxblox::Transformation::Position(T[4], T[5], T[6]));
        std::vector<voxblox::Point> pointcloud(pts.size() / 3);
        std::vector<voxblox::Color> colors(pts.size() / 3);
        for (size_t q = 0; q < pointcloud.size(); ++q) {
          pointcloud[q] = voxblox::Point(pts[3 * q], pts[3 * q + 1], pts[3 * q + 2]);
          colors[q].r = cols[4 * q]; colors[q].g = cols[4 * q + 1]; colors[q].b = cols[4 * q + 2]; colors[q].a = cols[4 * q + 3];
        }
        gpu.integratePointCloud(T_G_C, pointcloud, colors);
      }
      orc_tsdf_integrate(oi, T, pts.data(), cols.data(), (int64_t)pts.size() / 3, 0);
      ++tot.scans;
    }
    voxblox::Layer<voxblox::TsdfVoxel> host_layer(vs2, vps);
    voxgraph_amd::DownloadTsdfLayer(gpu_layer, &host_layer);
    compare(ol, host_layer, vps, &tot, "disjoint rays, no carving");
    growths += vgx_tsdf_layer_growths(gpu_layer.handle());
    int64_t dropped = -1;
    gpu_layer.getNumberOfAllocatedBlocks(&dropped);
    if (dropped != 0) {
      std::printf("FAIL: %lld dropped updates\n", (long long)dropped);
      return 1;
    }
    orc_tsdf_integrator_destroy(oi);
    orc_tsdf_layer_destroy(ol);
  }
  std::printf("TSDF_DROPIN scans=%lld voxels=%lld differing=%lld blocks_cpu=%lld blocks_gpu=%lld growths=%lld\n",
              tot.scans, tot.voxels, tot.differing, tot.blocks_cpu, tot.blocks_gpu, growths);
  vgx_ctx_destroy(ctx);
  return tot.differing == 0 ? 0 : 2;
}
