// voxblox::Layer: hashed block map ([recalled]).  TEST INFRASTRUCTURE.
#ifndef ORACLE_REF_SHIMS_VOXBLOX_CORE_LAYER_H_
#define ORACLE_REF_SHIMS_VOXBLOX_CORE_LAYER_H_
#include "voxblox/core/block.h"
namespace voxblox {
template <typename VoxelType>
class Layer {
 public:
  typedef Block<VoxelType> BlockType;
  typedef std::shared_ptr<Layer> Ptr;
  Layer(FloatingPoint voxel_size, size_t voxels_per_side)
      : voxel_size_(voxel_size), vps_(voxels_per_side) {
    voxel_size_inv_ = 1.0 / voxel_size_;
    block_size_ = voxel_size_ * vps_;
    block_size_inv_ = 1.0 / block_size_;
  }
  // deep copy, as voxblox's Layer copy constructor
  Layer(const Layer& o)
      : voxel_size_(o.voxel_size_), voxel_size_inv_(o.voxel_size_inv_), block_size_(o.block_size_),
        block_size_inv_(o.block_size_inv_), vps_(o.vps_) {
    for (const auto& kv : o.blocks_) blocks_.emplace(kv.first, std::make_shared<BlockType>(*kv.second));
  }
  bool hasBlock(const BlockIndex& index) const { return blocks_.count(index) > 0; }
  const BlockType& getBlockByIndex(const BlockIndex& index) const {
    auto it = blocks_.find(index);
    CHECK(it != blocks_.end()) << "block does not exist";
    return *it->second;
  }
  FloatingPoint voxel_size() const { return voxel_size_; }
  FloatingPoint voxel_size_inv() const { return voxel_size_inv_; }
  FloatingPoint block_size() const { return block_size_; }
  FloatingPoint block_size_inv() const { return block_size_inv_; }
  size_t voxels_per_side() const { return vps_; }
  BlockIndex computeBlockIndexFromCoordinates(const Point& coords) const {
    return getGridIndexFromPoint(coords, block_size_inv_);
  }
  typename BlockType::ConstPtr getBlockPtrByIndex(const BlockIndex& index) const {
    auto it = blocks_.find(index);
    return it == blocks_.end() ? typename BlockType::ConstPtr() : it->second;
  }
  typename BlockType::Ptr allocateBlockPtrByIndex(const BlockIndex& index) {
    auto it = blocks_.find(index);
    if (it != blocks_.end()) return it->second;
    Point origin(static_cast<FloatingPoint>(index[0]) * block_size_,
                 static_cast<FloatingPoint>(index[1]) * block_size_,
                 static_cast<FloatingPoint>(index[2]) * block_size_);
    auto block = std::make_shared<BlockType>(vps_, voxel_size_, origin);
    blocks_.emplace(index, block);
    return block;
  }
  size_t getNumberOfAllocatedBlocks() const { return blocks_.size(); }
  void getAllAllocatedBlocks(BlockIndexList* out) const {
    out->clear();
    for (const auto& kv : blocks_) out->push_back(kv.first);
  }

 private:
  FloatingPoint voxel_size_, voxel_size_inv_, block_size_, block_size_inv_;
  size_t vps_;
  std::unordered_map<BlockIndex, typename BlockType::Ptr, AnyIndexHash, AnyIndexEqual> blocks_;
};
}  // namespace voxblox
#endif
