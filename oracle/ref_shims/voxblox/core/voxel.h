// voxblox voxel payloads ([recalled]).  TEST INFRASTRUCTURE.
#ifndef ORACLE_REF_SHIMS_VOXBLOX_CORE_VOXEL_H_
#define ORACLE_REF_SHIMS_VOXBLOX_CORE_VOXEL_H_
#include "voxblox/core/common.h"
namespace voxblox {
struct TsdfVoxel {
  float distance = 0.0f;
  float weight = 0.0f;
  Color color;
};
struct EsdfVoxel {
  float distance = 0.0f;
  bool observed = false;
  bool hallucinated = false;
  bool in_queue = false;
  bool fixed = false;
  Eigen::Vector3i parent;
};
}  // namespace voxblox
#endif
