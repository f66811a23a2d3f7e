// The one-line swap at the reference's construction sites.
//
// voxgraph builds its registration cost functions from two VoxgraphSubmap::ConstPtr and a
// RegistrationCostFunction::Config (registration_constraint.cpp:33-35, submap_registration_helper.cpp:44-46,
// map_evaluation.cpp:143-144):
//     cost_function = new RegistrationCostFunction(config_.first_submap_ptr, config_.second_submap_ptr, config_.registration);
// With this header the GPU variant takes the SAME three arguments:
//     cost_function = voxgraph_amd::MakeGpuRegistrationCostFunction(config_.first_submap_ptr, config_.second_submap_ptr, config_.registration);
// The finished submaps' device copies live in a process-wide registry (a finished submap is immutable,
// voxgraph_mapper.cpp:464-471): uploaded on first use through voxgraph_submap_bridge.h; an entry is believed only while
// the shared_ptr that made it still owns the object at that address and the submap's stamp (ID, block counts, point
// counts) is unchanged -- otherwise it is uploaded again; released with GpuSubmapRegistry::release(submap), when its
// owner dies, or all at once with clear().
// oracle/ref_driver/callers_check.cpp compiles the reference's own registration_constraint.cpp and
// submap_registration_helper.cpp with exactly that edit (applied by sed at build time) and runs
// PoseGraph::optimize() both ways.
#ifndef VOXGRAPH_AMD_CPP_GPU_SUBMAP_REGISTRY_H_
#define VOXGRAPH_AMD_CPP_GPU_SUBMAP_REGISTRY_H_

#include <map>
#include <memory>
#include <mutex>
#include <stdexcept>

#include "gpu_registration_cost_function.h"
#include "voxgraph_submap_bridge.h"

namespace voxgraph_amd {

// What a cached device copy was made from, without reading a voxel: the submap's ID, the block counts of its two layers
// and the sizes of its two registration-point sets.  A submap that is finished again (voxgraph_submap.cpp:84-107) or
// edited in place changes at least one of them in practice; a caller that changes voxel VALUES behind a finished submap's
// back (the reference never does: voxgraph_mapper.cpp:464-471) calls release() itself.
struct GpuSubmapStamp {
  long long id = 0;
  size_t tsdf_blocks = 0, esdf_blocks = 0, voxel_points = 0, isosurface_points = 0;
  bool operator==(const GpuSubmapStamp& o) const {
    return id == o.id && tsdf_blocks == o.tsdf_blocks && esdf_blocks == o.esdf_blocks && voxel_points == o.voxel_points &&
           isosurface_points == o.isosurface_points;
  }
};
template <typename SubmapT>
GpuSubmapStamp StampOf(const SubmapT& submap) {
  GpuSubmapStamp s;
  s.id = static_cast<long long>(submap.getID());
  s.tsdf_blocks = submap.getTsdfMap().getTsdfLayer().getNumberOfAllocatedBlocks();
  s.esdf_blocks = submap.getEsdfMap().getEsdfLayer().getNumberOfAllocatedBlocks();
  using PointType = typename SubmapT::RegistrationPointType;
  s.voxel_points = submap.getRegistrationPoints(PointType::kVoxels).size();
  s.isosurface_points = submap.getRegistrationPoints(PointType::kIsosurfacePoints).size();
  return s;
}

class GpuSubmapRegistry {
 public:
  static GpuSubmapRegistry& instance() {
    static GpuSubmapRegistry r;
    return r;
  }
  // the context every cost function made through MakeGpuRegistrationCostFunction lives on (not owned)
  void setContext(vgx_ctx ctx) {
    std::lock_guard<std::mutex> lk(mu_);
    ctx_ = ctx;
  }
  vgx_ctx context() const {
    std::lock_guard<std::mutex> lk(mu_);
    return ctx_;
  }
  // The device copy of the submap behind `submap_ptr` (VoxgraphSubmap::ConstPtr, a std::shared_ptr), uploaded on first
  // use.  The cache is keyed by the object's address but BELIEVES an entry only while (a) the shared_ptr that made it
  // still owns the same object -- voxgraph dropping a submap without release() and the allocator handing the address to
  // the next one must not resurrect the old device copy -- and (b) the submap still carries the stamp it was
  // uploaded with.  Anything else is uploaded again, the stale copy destroyed.  Entries whose owner has died are swept
  // on every call.
  template <typename SubmapT>
  vgx_submap handleOf(const std::shared_ptr<SubmapT>& submap_ptr) {
    std::lock_guard<std::mutex> lk(mu_);
    if (!ctx_) throw std::runtime_error("GpuSubmapRegistry: setContext() first");
    if (!submap_ptr) throw std::runtime_error("GpuSubmapRegistry: null submap pointer");
    for (auto it = handles_.begin(); it != handles_.end();) {
      if (it->second.owner.expired()) {
        vgx_submap_destroy(it->second.handle);
        it = handles_.erase(it);
      } else {
        ++it;
      }
    }
    const void* key = static_cast<const void*>(submap_ptr.get());
    const GpuSubmapStamp stamp = StampOf(*submap_ptr);
    auto it = handles_.find(key);
    if (it != handles_.end()) {
      const std::weak_ptr<const void>& owner = it->second.owner;
      const bool same_owner = !owner.owner_before(submap_ptr) && !submap_ptr.owner_before(owner);
      if (same_owner && it->second.stamp == stamp) return it->second.handle;
      vgx_submap_destroy(it->second.handle);   // another object at the old address, or the same one finished again
      handles_.erase(it);
      ++stale_replaced_;
    }
    Entry e;
    e.handle = UploadFinishedSubmap(ctx_, *submap_ptr);
    e.owner = std::shared_ptr<const void>(submap_ptr);
    e.stamp = stamp;
    handles_[key] = e;
    return e.handle;
  }
  template <typename SubmapT>
  void release(const SubmapT& submap) {
    std::lock_guard<std::mutex> lk(mu_);
    auto it = handles_.find(static_cast<const void*>(&submap));
    if (it == handles_.end()) return;
    vgx_submap_destroy(it->second.handle);
    handles_.erase(it);
  }
  void clear() {
    std::lock_guard<std::mutex> lk(mu_);
    for (auto& kv : handles_) vgx_submap_destroy(kv.second.handle);
    handles_.clear();
  }
  size_t size() const {
    std::lock_guard<std::mutex> lk(mu_);
    return handles_.size();
  }
  long staleReplaced() const {
    std::lock_guard<std::mutex> lk(mu_);
    return stale_replaced_;
  }

 private:
  struct Entry {
    std::weak_ptr<const void> owner;
    vgx_submap handle = nullptr;
    GpuSubmapStamp stamp;
  };
  GpuSubmapRegistry() = default;
  mutable std::mutex mu_;
  vgx_ctx ctx_ = nullptr;
  std::map<const void*, Entry> handles_;
  long stale_replaced_ = 0;
};

// (reference submap pointer, reading submap pointer, RegistrationCostFunction::Config) -> ceres::CostFunction*
// config.jacobian_evaluation_method is not read here, as it is not read by RegistrationCostFunction either: the
// reference's CALL SITES wrap the cost function in a ceres::NumericDiffCostFunction when it says kNumeric
// (registration_constraint.cpp:28-32, submap_registration_helper.cpp:50-57), and they do so around this one just the same.
template <typename SubmapPtrT, typename RegistrationConfigT>
GpuRegistrationCostFunction* MakeGpuRegistrationCostFunction(const SubmapPtrT& reference_submap_ptr,
                                                             const SubmapPtrT& reading_submap_ptr,
                                                             const RegistrationConfigT& config) {
  GpuSubmapRegistry& registry = GpuSubmapRegistry::instance();
  GpuRegistrationCostFunction::Config gcfg;
  gcfg.registration_point_type = static_cast<int>(config.registration_point_type);
  gcfg.sampling_ratio = config.sampling_ratio;
  gcfg.no_correspondence_cost = config.no_correspondence_cost;
  gcfg.use_esdf_distance = config.use_esdf_distance;
  return new GpuRegistrationCostFunction(registry.context(), registry.handleOf(reference_submap_ptr),
                                         registry.handleOf(reading_submap_ptr), gcfg);
}

}  // namespace voxgraph_amd
#endif  // VOXGRAPH_AMD_CPP_GPU_SUBMAP_REGISTRY_H_
