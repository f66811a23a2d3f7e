// voxblox::Layer<TsdfVoxel>  <->  the GPU-resident layer of the active submap.
//
// The reference mutates the active submap's voxblox layer in place, scan after scan
// (voxgraph/src/frontend/measurement_processors/pointcloud_integrator.cpp:66-83), and
// finishSubmap() then reads that layer on the host (voxgraph_submap.cpp:84-107: generateEsdf(),
// findRelevantVoxelIndices(), findIsosurfaceVertices()).  With the GPU integrator the scans go into
// a GpuTsdfLayer instead; these two functions are the hand-over points:
//
//   // VoxgraphSubmap::finishSubmap(), before generateEsdf()                     voxgraph_submap.cpp:86
//   voxgraph_amd::DownloadTsdfLayer(*gpu_layer, tsdf_map_->getTsdfLayerPtr());
//
//   // taking over a submap that already holds data (a loaded map, or a CPU-integrated start)
//   voxgraph_amd::UploadTsdfLayer(submap.getTsdfMap().getTsdfLayer(), gpu_layer.get());
//
// Header-only, against voxblox's public API [recalled: Layer::allocateBlockPtrByIndex,
// getAllAllocatedBlocks, getBlockByIndex; Block::getVoxelByLinearIndex, num_voxels, has_data(),
// updated(); TsdfVoxel{distance, weight, color}; Color{r,g,b,a}]; exercised in this repository by
// the in-process TSDF drop-in check (tests/test_tsdf_dropin_gpu.py).
#ifndef VOXGRAPH_AMD_CPP_GPU_TSDF_LAYER_BRIDGE_H_
#define VOXGRAPH_AMD_CPP_GPU_TSDF_LAYER_BRIDGE_H_

#include <voxblox/core/layer.h>
#include <voxblox/core/voxel.h>

#include <stdexcept>
#include <string>
#include <vector>

#include "gpu_fast_tsdf_integrator.h"

namespace voxgraph_amd {

// GPU layer -> voxblox layer.  Every block the GPU layer holds is allocated in `layer` (if it is not
// there yet) and overwritten voxel for voxel; blocks only `layer` has are left alone.
inline void DownloadTsdfLayer(const GpuTsdfLayer& gpu, voxblox::Layer<voxblox::TsdfVoxel>* layer) {
  if (!layer) throw std::invalid_argument("DownloadTsdfLayer: layer == nullptr");
  if (static_cast<int>(layer->voxels_per_side()) != gpu.voxels_per_side())
    throw std::invalid_argument("DownloadTsdfLayer: voxels_per_side differs");
  int64_t dropped = 0;
  const int32_t n = gpu.getNumberOfAllocatedBlocks(&dropped);
  if (dropped != 0)
    throw std::runtime_error("DownloadTsdfLayer: the GPU layer dropped " + std::to_string(dropped) + " voxel updates");
  if (n == 0) return;
  const size_t vpb = static_cast<size_t>(gpu.voxels_per_side()) * gpu.voxels_per_side() * gpu.voxels_per_side();
  std::vector<int32_t> index(3 * static_cast<size_t>(n));
  std::vector<float> distance(vpb * n), weight(vpb * n);
  std::vector<uint8_t> rgba(4 * vpb * n);
  if (vgx_tsdf_layer_download(gpu.handle(), index.data(), distance.data(), weight.data(), rgba.data()) != VGX_OK)
    throw std::runtime_error("vgx_tsdf_layer_download failed");
  for (int32_t b = 0; b < n; ++b) {
    voxblox::BlockIndex bi;
    bi[0] = index[3 * b];
    bi[1] = index[3 * b + 1];
    bi[2] = index[3 * b + 2];
    auto block = layer->allocateBlockPtrByIndex(bi);
    for (size_t lin = 0; lin < vpb; ++lin) {
      voxblox::TsdfVoxel& v = block->getVoxelByLinearIndex(lin);
      const size_t at = static_cast<size_t>(b) * vpb + lin;
      v.distance = distance[at];
      v.weight = weight[at];
      v.color.r = rgba[4 * at];
      v.color.g = rgba[4 * at + 1];
      v.color.b = rgba[4 * at + 2];
      v.color.a = rgba[4 * at + 3];
    }
    // what TsdfIntegratorBase::allocateStorageAndGetVoxelPtr / updateLayerWithStoredBlocks leave
    // behind on every block they touched [recalled]: downstream ESDF / mesh integrators look at these
    block->has_data() = true;
    block->updated().set();
  }
}

// voxblox layer -> GPU layer (replaces the GPU layer's contents).
inline void UploadTsdfLayer(const voxblox::Layer<voxblox::TsdfVoxel>& layer, GpuTsdfLayer* gpu) {
  if (!gpu) throw std::invalid_argument("UploadTsdfLayer: gpu == nullptr");
  if (static_cast<int>(layer.voxels_per_side()) != gpu->voxels_per_side())
    throw std::invalid_argument("UploadTsdfLayer: voxels_per_side differs");
  voxblox::BlockIndexList blocks;
  layer.getAllAllocatedBlocks(&blocks);
  const size_t n = blocks.size();
  const size_t vpb = static_cast<size_t>(gpu->voxels_per_side()) * gpu->voxels_per_side() * gpu->voxels_per_side();
  std::vector<int32_t> index(3 * n);
  std::vector<float> distance(vpb * n), weight(vpb * n);
  std::vector<uint8_t> rgba(4 * vpb * n);
  for (size_t b = 0; b < n; ++b) {
    for (int a = 0; a < 3; ++a) index[3 * b + a] = blocks[b][a];
    const auto& block = layer.getBlockByIndex(blocks[b]);
    for (size_t lin = 0; lin < vpb; ++lin) {
      const voxblox::TsdfVoxel& v = block.getVoxelByLinearIndex(lin);
      const size_t at = b * vpb + lin;
      distance[at] = v.distance;
      weight[at] = v.weight;
      rgba[4 * at] = v.color.r;
      rgba[4 * at + 1] = v.color.g;
      rgba[4 * at + 2] = v.color.b;
      rgba[4 * at + 3] = v.color.a;
    }
  }
  if (vgx_tsdf_layer_upload(gpu->handle(), static_cast<int32_t>(n), index.data(), distance.data(), weight.data(),
                            rgba.data()) != VGX_OK)
    throw std::runtime_error(std::string("vgx_tsdf_layer_upload: ") + gpu->last_error());
}

}  // namespace voxgraph_amd

#endif  // VOXGRAPH_AMD_CPP_GPU_TSDF_LAYER_BRIDGE_H_
