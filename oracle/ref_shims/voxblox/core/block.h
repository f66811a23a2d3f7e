// voxblox::Block geometry ([recalled]).  TEST INFRASTRUCTURE.
#ifndef ORACLE_REF_SHIMS_VOXBLOX_CORE_BLOCK_H_
#define ORACLE_REF_SHIMS_VOXBLOX_CORE_BLOCK_H_
#include <algorithm>
#include <bitset>

#include "voxblox/core/voxel.h"
namespace voxblox {
template <typename VoxelType>
class Block {
 public:
  typedef std::shared_ptr<Block<VoxelType>> Ptr;
  typedef std::shared_ptr<const Block<VoxelType>> ConstPtr;
  Block(size_t voxels_per_side, FloatingPoint voxel_size, const Point& origin)
      : vps_(voxels_per_side), voxel_size_(voxel_size), voxel_size_inv_(1.0 / voxel_size),
        origin_(origin), voxels_(voxels_per_side * voxels_per_side * voxels_per_side) {}
  size_t voxels_per_side() const { return vps_; }
  FloatingPoint voxel_size() const { return voxel_size_; }
  FloatingPoint voxel_size_inv() const { return voxel_size_inv_; }
  const Point& origin() const { return origin_; }
  size_t num_voxels() const { return voxels_.size(); }
  size_t computeLinearIndexFromVoxelIndex(const VoxelIndex& i) const {
    return static_cast<size_t>(i[0] + vps_ * (i[1] + i[2] * vps_));
  }
  VoxelIndex computeTruncatedVoxelIndexFromCoordinates(const Point& coords) const {
    const int max_value = static_cast<int>(vps_) - 1;
    VoxelIndex idx = getGridIndexFromPoint(coords - origin_, voxel_size_inv_);
    for (int a = 0; a < 3; ++a) idx[a] = std::max(std::min(idx[a], max_value), 0);
    return idx;
  }
  Point computeCoordinatesFromVoxelIndex(const VoxelIndex& idx) const {
    return origin_ + getCenterPointFromGridIndex(idx, voxel_size_);
  }
  Point computeCoordinatesFromLinearIndex(size_t lin) const {
    VoxelIndex idx;
    idx[0] = static_cast<int>(lin % vps_);
    idx[1] = static_cast<int>((lin / vps_) % vps_);
    idx[2] = static_cast<int>(lin / (vps_ * vps_));
    return computeCoordinatesFromVoxelIndex(idx);
  }
  const VoxelType& getVoxelByLinearIndex(size_t i) const { return voxels_[i]; }
  VoxelType& getVoxelByLinearIndex(size_t i) { return voxels_[i]; }
  const VoxelType& getVoxelByVoxelIndex(const VoxelIndex& i) const {
    return voxels_[computeLinearIndexFromVoxelIndex(i)];
  }
  bool isValidLinearIndex(size_t i) const { return i < voxels_.size(); }
  // bookkeeping flags of voxblox::Block [recalled]: Update::{kMap, kMesh, kEsdf}
  bool has_data() const { return has_data_; }
  bool& has_data() { return has_data_; }
  const std::bitset<3>& updated() const { return updated_; }
  std::bitset<3>& updated() { return updated_; }

 private:
  size_t vps_;
  FloatingPoint voxel_size_, voxel_size_inv_;
  Point origin_;
  std::vector<VoxelType> voxels_;
  bool has_data_ = false;
  std::bitset<3> updated_;
};
}  // namespace voxblox
#endif
