// voxblox basic types ([recalled]).  TEST INFRASTRUCTURE.
#ifndef ORACLE_REF_SHIMS_VOXBLOX_CORE_COMMON_H_
#define ORACLE_REF_SHIMS_VOXBLOX_CORE_COMMON_H_
#include <cmath>
#include <cstdint>
#include <iostream>
#include <memory>
#include <unordered_map>
#include <unordered_set>
#include <vector>

#include <Eigen/Core>
#include <glog/logging.h>
#include <kindr/minimal/quat-transformation.h>

namespace voxblox {
typedef float FloatingPoint;
typedef Eigen::Matrix<FloatingPoint, 3, 1> Point;
typedef Eigen::Matrix<int, 3, 1> AnyIndex;
typedef AnyIndex BlockIndex;
typedef AnyIndex VoxelIndex;
typedef std::vector<BlockIndex> BlockIndexList;
typedef Eigen::Matrix<FloatingPoint, 1, 8> InterpVector;
typedef Eigen::Matrix<FloatingPoint, 8, 8> InterpTable;
typedef kindr::minimal::QuatTransformationTemplate<FloatingPoint> Transformation;

constexpr FloatingPoint kCoordinateEpsilon = 1e-6;

struct Color {
  uint8_t r = 0, g = 0, b = 0, a = 0;
  Color() {}
  Color(uint8_t r_, uint8_t g_, uint8_t b_, uint8_t a_ = 255) : r(r_), g(g_), b(b_), a(a_) {}
};

// floor(p * inv + eps) per axis
inline AnyIndex getGridIndexFromPoint(const Point& p, const FloatingPoint grid_size_inv) {
  AnyIndex idx;
  for (int a = 0; a < 3; ++a) idx[a] = static_cast<int>(std::floor(p[a] * grid_size_inv + kCoordinateEpsilon));
  return idx;
}
template <typename IndexType>
inline IndexType getGridIndexFromPoint(const Point& p, const FloatingPoint grid_size_inv) {
  return getGridIndexFromPoint(p, grid_size_inv);
}
// (idx + 0.5) * grid_size: the sum and product are formed in double (the 0.5 literal)
inline Point getCenterPointFromGridIndex(const AnyIndex& idx, FloatingPoint grid_size) {
  return Point(static_cast<FloatingPoint>((static_cast<FloatingPoint>(idx[0]) + 0.5) * grid_size),
               static_cast<FloatingPoint>((static_cast<FloatingPoint>(idx[1]) + 0.5) * grid_size),
               static_cast<FloatingPoint>((static_cast<FloatingPoint>(idx[2]) + 0.5) * grid_size));
}

struct AnyIndexHash {
  static constexpr size_t sl = 17191;
  static constexpr size_t sl2 = sl * sl;
  size_t operator()(const AnyIndex& i) const {
    return static_cast<unsigned int>(i[0] + i[1] * sl + i[2] * sl2);
  }
};
struct AnyIndexEqual {
  bool operator()(const AnyIndex& a, const AnyIndex& b) const {
    return a[0] == b[0] && a[1] == b[1] && a[2] == b[2];
  }
};
typedef std::unordered_set<AnyIndex, AnyIndexHash, AnyIndexEqual> IndexSet;
template <typename T>
using AlignedVector = std::vector<T>;
}  // namespace voxblox
#endif
