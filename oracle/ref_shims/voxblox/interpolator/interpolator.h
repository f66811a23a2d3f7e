// voxblox::Interpolator::getVoxelsAndQVector ([recalled]).
// TEST INFRASTRUCTURE -- see oracle/ref_shims/README.md.
#ifndef ORACLE_REF_SHIMS_VOXBLOX_INTERPOLATOR_INTERPOLATOR_H_
#define ORACLE_REF_SHIMS_VOXBLOX_INTERPOLATOR_INTERPOLATOR_H_
#include "voxblox/core/layer.h"
namespace voxblox {

inline bool isVoxelValidForInterpolation(const TsdfVoxel& v) { return v.weight > 0.0f; }
inline bool isVoxelValidForInterpolation(const EsdfVoxel& v) { return v.observed; }

template <typename VoxelType>
class Interpolator {
 public:
  explicit Interpolator(const Layer<VoxelType>* layer) : layer_(layer) {}

  // getVoxel(pos, &voxel, interpolate = true): every field trilinearly interpolated with the
  // same q-vector / B_1 table; without interpolation, the voxel containing pos
  bool getVoxel(const Point& pos, VoxelType* voxel, bool interpolate = false) const {
    if (!interpolate) {
      typename Layer<VoxelType>::BlockType::ConstPtr block =
          layer_->getBlockPtrByIndex(layer_->computeBlockIndexFromCoordinates(pos));
      if (!block) return false;
      *voxel = block->getVoxelByVoxelIndex(block->computeTruncatedVoxelIndexFromCoordinates(pos));
      return true;
    }
    const VoxelType* voxels[8];
    InterpVector q;
    if (!getVoxelsAndQVector(pos, voxels, &q)) return false;
    *voxel = interpVoxel(q, voxels);
    return true;
  }

  // the eight voxels surrounding `pos` (neighbour k sits at base + (k>>2&1, k>>1&1, k&1))
  // and q = [1, dx, dy, dz, dx dy, dy dz, dz dx, dx dy dz] relative to the base voxel's centre;
  // false as soon as a block is missing or a voxel is not valid
  bool getVoxelsAndQVector(const Point& pos, const VoxelType** voxels, InterpVector* q_vector) const {
    BlockIndex block_index;
    VoxelIndex base;
    if (!setIndexes(pos, &block_index, &base)) return false;
    const int vps = static_cast<int>(layer_->voxels_per_side());
    for (int k = 0; k < 8; ++k) {
      typename Layer<VoxelType>::BlockType::ConstPtr block = layer_->getBlockPtrByIndex(block_index);
      if (!block) return false;
      VoxelIndex v;
      v[0] = base[0] + ((k >> 2) & 1);
      v[1] = base[1] + ((k >> 1) & 1);
      v[2] = base[2] + (k & 1);
      if (v[0] >= vps || v[1] >= vps || v[2] >= vps) {
        BlockIndex shifted = block_index;
        for (int a = 0; a < 3; ++a)
          if (v[a] >= vps) {
            shifted[a]++;
            v[a] -= vps;
          }
        block = layer_->getBlockPtrByIndex(shifted);
        if (!block) return false;
      }
      if (k == 0) {
        const Point centre = block->computeCoordinatesFromVoxelIndex(v);
        const Point d = (pos - centre) * block->voxel_size_inv();
        *q_vector << 1, d[0], d[1], d[2], d[0] * d[1], d[1] * d[2], d[2] * d[0], d[0] * d[1] * d[2];
      }
      const VoxelType& voxel = block->getVoxelByVoxelIndex(v);
      voxels[k] = &voxel;
      if (!isVoxelValidForInterpolation(voxel)) return false;
    }
    return true;
  }

 private:
  static InterpTable table() {
    return (InterpTable() << 1, 0, 0, 0, 0, 0, 0, 0,    -1, 0, 0, 0, 1, 0, 0, 0,   -1, 0, 1, 0, 0, 0, 0, 0,
            -1, 1, 0, 0, 0, 0, 0, 0,   1, 0, -1, 0, -1, 0, 1, 0,   1, -1, -1, 1, 0, 0, 0, 0,
            1, -1, 0, 0, -1, 1, 0, 0,  -1, 1, 1, -1, 1, -1, -1, 1).finished();
  }
  static TsdfVoxel interpVoxel(const InterpVector& q, const TsdfVoxel** voxels) {
    InterpVector d, w;
    for (int i = 0; i < 8; ++i) {
      d[i] = voxels[i]->distance;
      w[i] = voxels[i]->weight;
    }
    TsdfVoxel out;
    out.distance = q * (table() * d.transpose());
    out.weight = q * (table() * w.transpose());
    return out;
  }
  static EsdfVoxel interpVoxel(const InterpVector& q, const EsdfVoxel** voxels) {
    InterpVector d;
    for (int i = 0; i < 8; ++i) d[i] = voxels[i]->distance;
    EsdfVoxel out;
    out.distance = q * (table() * d.transpose());
    out.observed = true;
    return out;
  }
  // block containing pos (must exist) and the voxel whose centre is the lower corner of the
  // interpolation cell; stepping below index 0 moves to the previous block
  bool setIndexes(const Point& pos, BlockIndex* block_index, VoxelIndex* base) const {
    *block_index = layer_->computeBlockIndexFromCoordinates(pos);
    typename Layer<VoxelType>::BlockType::ConstPtr block = layer_->getBlockPtrByIndex(*block_index);
    if (!block) return false;
    VoxelIndex v = block->computeTruncatedVoxelIndexFromCoordinates(pos);
    const Point centre_offset = pos - block->computeCoordinatesFromVoxelIndex(v);
    const int vps = static_cast<int>(block->voxels_per_side());
    for (int a = 0; a < 3; ++a) {
      if (centre_offset[a] < 0) {
        v[a]--;
        if (v[a] < 0) {
          (*block_index)[a]--;
          v[a] += vps;
        }
      }
    }
    *base = v;
    return true;
  }
  const Layer<VoxelType>* layer_;
};
}  // namespace voxblox
#endif
