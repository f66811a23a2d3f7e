// Host-side mirror of voxblox::FastTsdfIntegrator as voxgraph drives it
// (voxgraph/src/frontend/measurement_processors/pointcloud_integrator.cpp:66-83):
//   tsdf_integrator_.reset(new voxblox::FastTsdfIntegrator(config, layer_ptr));   :67-69
//   tsdf_integrator_->setLayer(layer_ptr);                                         :77
//   tsdf_integrator_->integratePointCloud(T_submap_sensor, pointcloud, colors);    :83
// No voxblox headers are needed: points are AoS float3 (voxblox::Pointcloud is a
// std::vector<Eigen::Vector3f>, contiguous 12-byte elements), colours AoS RGBA8
// (voxblox::Color), the transform is {qw,qx,qy,qz,tx,ty,tz} of voxblox::Transformation.
#ifndef VOXGRAPH_AMD_CPP_GPU_FAST_TSDF_INTEGRATOR_H_
#define VOXGRAPH_AMD_CPP_GPU_FAST_TSDF_INTEGRATOR_H_

#include <cstdint>
#include <stdexcept>
#include <string>
#include <vector>

#include "voxgraph_amd.h"

namespace voxgraph_amd {

// The active submap's TSDF layer on the GPU: voxblox::Layer<TsdfVoxel>(voxel_size, voxels_per_side),
// unbounded like it.  gpu_tsdf_layer_bridge.h moves its contents to / from a voxblox layer.
class GpuTsdfLayer {
 public:
  GpuTsdfLayer(vgx_ctx ctx, float voxel_size, int voxels_per_side) : ctx_(ctx), vps_(voxels_per_side), voxel_size_(voxel_size) {
    if (vgx_tsdf_layer_create(ctx, voxel_size, voxels_per_side, nullptr, nullptr, 0, &layer_) != VGX_OK)
      throw std::runtime_error(std::string("vgx_tsdf_layer_create: ") + vgx_last_error(ctx));
  }
  // with an initial reservation (block box and pool size); the layer still grows beyond it
  GpuTsdfLayer(vgx_ctx ctx, float voxel_size, int voxels_per_side, const int32_t box_min[3],
               const int32_t box_dim[3], int32_t max_blocks)
      : ctx_(ctx), vps_(voxels_per_side), voxel_size_(voxel_size) {
    if (vgx_tsdf_layer_create(ctx, voxel_size, voxels_per_side, box_min, box_dim, max_blocks,
                              &layer_) != VGX_OK)
      throw std::runtime_error(std::string("vgx_tsdf_layer_create: ") + vgx_last_error(ctx));
  }
  ~GpuTsdfLayer() { vgx_tsdf_layer_destroy(layer_); }
  GpuTsdfLayer(const GpuTsdfLayer&) = delete;
  GpuTsdfLayer& operator=(const GpuTsdfLayer&) = delete;
  vgx_tsdf_layer handle() const { return layer_; }
  int voxels_per_side() const { return vps_; }
  float voxel_size() const { return voxel_size_; }
  const char* last_error() const { return vgx_last_error(ctx_); }
  // waits for the scans in flight; *dropped_updates (nullable) is 0 unless the GPU ran out of memory
  int32_t getNumberOfAllocatedBlocks(int64_t* dropped_updates = nullptr) const {
    int32_t n = 0;
    if (vgx_tsdf_layer_stats(layer_, &n, dropped_updates) != VGX_OK)
      throw std::runtime_error(std::string("vgx_tsdf_layer_stats: ") + vgx_last_error(ctx_));
    return n;
  }

 private:
  vgx_ctx ctx_;
  int vps_;
  float voxel_size_;
  vgx_tsdf_layer layer_ = nullptr;
};

class GpuFastTsdfIntegrator {
 public:
  // voxblox::TsdfIntegratorBase::Config, field for field where it matters on a GPU
  using Config = vgx_tsdf_config;
  static Config defaultConfig() {
    Config c;
    vgx_tsdf_config_default(&c);
    return c;
  }

  GpuFastTsdfIntegrator(vgx_ctx ctx, const Config& config, GpuTsdfLayer* layer) : ctx_(ctx) {
    if (vgx_tsdf_integrator_create(ctx, &config, layer ? layer->handle() : nullptr, &integ_) != VGX_OK)
      throw std::runtime_error(std::string("vgx_tsdf_integrator_create: ") + vgx_last_error(ctx));
  }
  ~GpuFastTsdfIntegrator() { vgx_tsdf_integrator_destroy(integ_); }
  GpuFastTsdfIntegrator(const GpuFastTsdfIntegrator&) = delete;
  GpuFastTsdfIntegrator& operator=(const GpuFastTsdfIntegrator&) = delete;

  void setLayer(GpuTsdfLayer* layer) {
    if (vgx_tsdf_integrator_set_layer(integ_, layer->handle()) != VGX_OK)
      throw std::runtime_error("vgx_tsdf_integrator_set_layer failed");
  }

  // integratePointCloud(T_G_C, points_C, colors, freespace_points = false)
  void integratePointCloud(const float T_G_C[7], const float* points_C, const uint8_t* colors,
                           int64_t n_points, bool freespace_points = false) {
    if (vgx_tsdf_integrate(integ_, T_G_C, points_C, colors, n_points, freespace_points ? 1 : 0,
                           nullptr) != VGX_OK)
      throw std::runtime_error(std::string("vgx_tsdf_integrate: ") + vgx_last_error(ctx_));
  }

  // The call voxgraph makes, with voxblox's own types (pointcloud_integrator.cpp:83, member type
  // pointcloud_integrator.h:30):   integratePointCloud(T_submap_sensor, pointcloud, colors, freespace_points = false)
  //   Transformation  kindr::minimal::QuatTransformationTemplate<float>: getRotation().{w,x,y,z}(), getPosition()[k]
  //   Pointcloud      AlignedVector<Eigen::Vector3f>: contiguous 12-byte elements
  //   Colors          AlignedVector<voxblox::Color>:  contiguous RGBA8; may be empty
  // Templated (containers only: arrays take the overload above), so it compiles against the real headers and against
  // header stand-ins (the checker tree has a set) alike.
  template <class Transformation, class Pointcloud, class Colors, class = typename Pointcloud::value_type,
            class = typename Colors::value_type>
  void integratePointCloud(const Transformation& T_G_C, const Pointcloud& points_C, const Colors& colors,
                           bool freespace_points = false) {
    static_assert(sizeof(typename Pointcloud::value_type) == 12, "Pointcloud: three packed floats per point");
    static_assert(sizeof(typename Colors::value_type) == 4, "Colors: RGBA8 per point");
    if (!colors.empty() && colors.size() != points_C.size())
      throw std::invalid_argument("integratePointCloud: colors and points differ in length");  // voxblox CHECK_EQ
    const auto& q = T_G_C.getRotation();
    const auto& t = T_G_C.getPosition();
    const float T[7] = {(float)q.w(), (float)q.x(), (float)q.y(), (float)q.z(), (float)t[0], (float)t[1], (float)t[2]};
    integratePointCloud(T, reinterpret_cast<const float*>(points_C.data()),
                        colors.empty() ? nullptr : reinterpret_cast<const uint8_t*>(colors.data()), (int64_t)points_C.size(),
                        freespace_points);
  }
  // an ORGANISED cloud's row length (sensor_msgs/PointCloud2.width, which voxgraph's callback has and voxblox's flat
  // Pointcloud drops): vgx_tsdf_integrator_set_cloud_width; 0 = unorganised
  void setCloudWidth(int32_t width) {
    if (vgx_tsdf_integrator_set_cloud_width(integ_, width) != VGX_OK) throw std::invalid_argument("setCloudWidth: negative width");
  }

 private:
  vgx_ctx ctx_;
  vgx_tsdf_integrator integ_ = nullptr;
};

// voxblox::MergedTsdfIntegrator (the variant north_star also names; voxgraph constructs the Fast
// one): same constructor, setLayer and integratePointCloud; points that end in the same voxel are
// merged into one ray.  Config::enable_anti_grazing is voxblox's option of that name.
class GpuMergedTsdfIntegrator {
 public:
  using Config = vgx_tsdf_config;
  GpuMergedTsdfIntegrator(vgx_ctx ctx, const Config& config, GpuTsdfLayer* layer) : ctx_(ctx) {
    if (vgx_tsdf_integrator_create(ctx, &config, layer ? layer->handle() : nullptr, &integ_) != VGX_OK)
      throw std::runtime_error(std::string("vgx_tsdf_integrator_create: ") + vgx_last_error(ctx));
  }
  ~GpuMergedTsdfIntegrator() { vgx_tsdf_integrator_destroy(integ_); }
  GpuMergedTsdfIntegrator(const GpuMergedTsdfIntegrator&) = delete;
  GpuMergedTsdfIntegrator& operator=(const GpuMergedTsdfIntegrator&) = delete;

  void setLayer(GpuTsdfLayer* layer) {
    if (vgx_tsdf_integrator_set_layer(integ_, layer->handle()) != VGX_OK)
      throw std::runtime_error("vgx_tsdf_integrator_set_layer failed");
  }

  void integratePointCloud(const float T_G_C[7], const float* points_C, const uint8_t* colors,
                           int64_t n_points, bool freespace_points = false) {
    if (vgx_tsdf_integrate_merged(integ_, T_G_C, points_C, colors, n_points, freespace_points ? 1 : 0,
                                  nullptr) != VGX_OK)
      throw std::runtime_error(std::string("vgx_tsdf_integrate_merged: ") + vgx_last_error(ctx_));
  }

  // The call voxgraph makes, with voxblox's own types (pointcloud_integrator.cpp:83, member type
  // pointcloud_integrator.h:30):   integratePointCloud(T_submap_sensor, pointcloud, colors, freespace_points = false)
  //   Transformation  kindr::minimal::QuatTransformationTemplate<float>: getRotation().{w,x,y,z}(), getPosition()[k]
  //   Pointcloud      AlignedVector<Eigen::Vector3f>: contiguous 12-byte elements
  //   Colors          AlignedVector<voxblox::Color>:  contiguous RGBA8; may be empty
  // Templated (containers only: arrays take the overload above), so it compiles against the real headers and against
  // header stand-ins (the checker tree has a set) alike.
  template <class Transformation, class Pointcloud, class Colors, class = typename Pointcloud::value_type,
            class = typename Colors::value_type>
  void integratePointCloud(const Transformation& T_G_C, const Pointcloud& points_C, const Colors& colors,
                           bool freespace_points = false) {
    static_assert(sizeof(typename Pointcloud::value_type) == 12, "Pointcloud: three packed floats per point");
    static_assert(sizeof(typename Colors::value_type) == 4, "Colors: RGBA8 per point");
    if (!colors.empty() && colors.size() != points_C.size())
      throw std::invalid_argument("integratePointCloud: colors and points differ in length");  // voxblox CHECK_EQ
    const auto& q = T_G_C.getRotation();
    const auto& t = T_G_C.getPosition();
    const float T[7] = {(float)q.w(), (float)q.x(), (float)q.y(), (float)q.z(), (float)t[0], (float)t[1], (float)t[2]};
    integratePointCloud(T, reinterpret_cast<const float*>(points_C.data()),
                        colors.empty() ? nullptr : reinterpret_cast<const uint8_t*>(colors.data()), (int64_t)points_C.size(),
                        freespace_points);
  }
  // (the merged integrator ignores the hint) an ORGANISED cloud's row length (sensor_msgs/PointCloud2.width, which voxgraph's callback has and voxblox's flat
  // Pointcloud drops): vgx_tsdf_integrator_set_cloud_width; 0 = unorganised
  void setCloudWidth(int32_t width) {
    if (vgx_tsdf_integrator_set_cloud_width(integ_, width) != VGX_OK) throw std::invalid_argument("setCloudWidth: negative width");
  }

 private:
  vgx_ctx ctx_;
  vgx_tsdf_integrator integ_ = nullptr;
};

}  // namespace voxgraph_amd

#endif  // VOXGRAPH_AMD_CPP_GPU_FAST_TSDF_INTEGRATOR_H_
