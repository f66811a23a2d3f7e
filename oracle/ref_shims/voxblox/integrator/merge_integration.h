// transformLayer (layer resampling) is only reached from VoxgraphSubmap::transformSubmap, which
// nothing on the hot path calls: declared so the translation unit links, aborts if used.
#ifndef ORACLE_REF_SHIMS_VOXBLOX_INTEGRATOR_MERGE_INTEGRATION_H_
#define ORACLE_REF_SHIMS_VOXBLOX_INTEGRATOR_MERGE_INTEGRATION_H_
#include "voxblox/core/layer.h"
namespace voxblox {
template <typename VoxelType>
void transformLayer(const Layer<VoxelType>&, const Transformation&, Layer<VoxelType>*) {
  CHECK(false) << "transformLayer is not part of the stand-in";
}
}  // namespace voxblox
#endif
