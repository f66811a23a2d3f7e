// Stand-in for voxblox's MeshIntegrator / MeshLayer / Mesh ([recalled], see oracle/iso_oracle.h
// for the statement of what is and is not reproduced): the mesh is reduced to its VERTEX SET --
// one vertex per sign-changing edge of the TSDF's dual grid that belongs to at least one cell
// whose 8 corners all have weight > min_weight (every marching-cubes case uses all of its
// sign-changing edges), interpolated low -> high along the edge.  getConnectedMesh keeps the
// first vertex met in each cell of size `threshold`.  Triangles are not built (voxgraph reads
// vertices only, voxgraph_submap.cpp:224).  Iteration order: getAllAllocatedBlocks order, voxel
// linear index, axis.  TEST INFRASTRUCTURE.
#ifndef ORACLE_REF_SHIMS_VOXBLOX_MESH_MESH_INTEGRATOR_H_
#define ORACLE_REF_SHIMS_VOXBLOX_MESH_MESH_INTEGRATOR_H_
#include <cmath>
#include <set>
#include <tuple>

#include "voxblox/core/layer.h"
namespace voxblox {

struct MeshIntegratorConfig {
  bool use_color = true;
  float min_weight = 1e-4;
};

class Mesh {
 public:
  Mesh(FloatingPoint block_size, const Point& origin) : block_size(block_size), origin(origin) {}
  AlignedVector<Point> vertices;
  FloatingPoint block_size;
  Point origin;
};

class MeshLayer {
 public:
  explicit MeshLayer(FloatingPoint block_size) : block_size_(block_size) {}
  void getConnectedMesh(Mesh* out, const FloatingPoint approximate_vertex_proximity_threshold) const {
    const double inv = 1.0 / static_cast<double>(approximate_vertex_proximity_threshold);
    std::set<std::tuple<long long, long long, long long>> taken;
    out->vertices.clear();
    for (const Point& v : raw_vertices) {
      auto key = std::make_tuple(static_cast<long long>(std::round(v[0] * inv)),
                                 static_cast<long long>(std::round(v[1] * inv)),
                                 static_cast<long long>(std::round(v[2] * inv)));
      if (taken.insert(key).second) out->vertices.push_back(v);
    }
  }
  AlignedVector<Point> raw_vertices;

 private:
  FloatingPoint block_size_;
};

template <typename VoxelType>
class MeshIntegrator {
 public:
  MeshIntegrator(const MeshIntegratorConfig& config, const Layer<VoxelType>& sdf_layer, MeshLayer* mesh_layer)
      : config_(config), layer_(sdf_layer), mesh_(mesh_layer) {}

  void generateMesh(bool /*only_mesh_updated_blocks*/, bool /*clear_updated_flag*/) {
    mesh_->raw_vertices.clear();
    const int vps = static_cast<int>(layer_.voxels_per_side());
    BlockIndexList blocks;
    layer_.getAllAllocatedBlocks(&blocks);
    for (const BlockIndex& bi : blocks) {
      const Block<VoxelType>& block = layer_.getBlockByIndex(bi);
      for (size_t lin = 0; lin < block.num_voxels(); ++lin) {
        const VoxelType& v0 = block.getVoxelByLinearIndex(lin);
        if (!(v0.weight > config_.min_weight)) continue;
        VoxelIndex vi;
        vi[0] = static_cast<int>(lin % vps);
        vi[1] = static_cast<int>((lin / vps) % vps);
        vi[2] = static_cast<int>(lin / (static_cast<size_t>(vps) * vps));
        for (int axis = 0; axis < 3; ++axis) {
          VoxelIndex step;
          step[axis] = 1;
          const VoxelType* v1 = at(bi, vi[0] + step[0], vi[1] + step[1], vi[2] + step[2]);
          if (!v1 || !(v1->weight > config_.min_weight)) continue;
          if ((v0.distance < 0.0f) == (v1->distance < 0.0f)) continue;
          if (!edgeHasObservedCell(bi, vi, axis)) continue;
          const Point p0 = block.computeCoordinatesFromVoxelIndex(vi);
          Point p1 = p0;
          {
            int nv = vi[axis] + 1, nb = bi[axis];
            if (nv >= vps) {
              nv -= vps;
              nb++;
            }
            p1[axis] = static_cast<FloatingPoint>(nb) * layer_.block_size() +
                       static_cast<FloatingPoint>((static_cast<FloatingPoint>(nv) + 0.5) * layer_.voxel_size());
          }
          Point vert = p0;
          const float sdf_diff = v0.distance - v1->distance;
          if (std::fabs(sdf_diff) >= 1e-6f) {
            const float t = v0.distance / sdf_diff;
            vert[axis] = p0[axis] + t * (p1[axis] - p0[axis]);
          } else {
            vert[axis] = 0.5f * (p0[axis] + p1[axis]);
          }
          mesh_->raw_vertices.push_back(vert);
        }
      }
    }
  }

 private:
  const VoxelType* at(const BlockIndex& b, int x, int y, int z) const {
    const int vps = static_cast<int>(layer_.voxels_per_side());
    BlockIndex bb = b;
    VoxelIndex v;
    v[0] = x;
    v[1] = y;
    v[2] = z;
    for (int a = 0; a < 3; ++a) {
      if (v[a] < 0) {
        v[a] += vps;
        bb[a]--;
      }
      if (v[a] >= vps) {
        v[a] -= vps;
        bb[a]++;
      }
    }
    typename Layer<VoxelType>::BlockType::ConstPtr block = layer_.getBlockPtrByIndex(bb);
    return block ? &block->getVoxelByVoxelIndex(v) : nullptr;
  }
  bool edgeHasObservedCell(const BlockIndex& b, const VoxelIndex& v, int axis) const {
    const int ob = (axis + 1) % 3, oc = (axis + 2) % 3;
    for (int sb = 0; sb < 2; ++sb)
      for (int sc = 0; sc < 2; ++sc) {
        int base[3] = {v[0], v[1], v[2]};
        base[ob] -= sb;
        base[oc] -= sc;
        bool all = true;
        for (int k = 0; k < 8 && all; ++k) {
          const VoxelType* c = at(b, base[0] + (k & 1), base[1] + ((k >> 1) & 1), base[2] + ((k >> 2) & 1));
          all = c && c->weight > config_.min_weight;
        }
        if (all) return true;
      }
    return false;
  }
  MeshIntegratorConfig config_;
  const Layer<VoxelType>& layer_;
  MeshLayer* mesh_;
};
}  // namespace voxblox
#endif
