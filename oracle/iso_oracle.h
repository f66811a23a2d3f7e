/*
 * oracle/iso_oracle.h -- CPU restatement of VoxgraphSubmap::findIsosurfaceVertices
 * (voxgraph/src/frontend/submap_collection/voxgraph_submap.cpp:203-243): the
 * kIsosurfacePoints registration points the shipped config registers with
 * ("explicit_to_implicit", voxgraph/config/voxgraph_mapper.yaml:35).
 * TEST INFRASTRUCTURE ONLY (usage rule in reg_oracle.h).
 *
 * PARITY: the part voxgraph owns (interpolated distance/weight per vertex, CHECK_LE, block set) is
 * pinned to the reference's own voxgraph_submap.cpp (tests/test_ref_submap_pin.py, oracle/_ref);
 * the vertex positions are UNPINNED.  The reference builds these points from three voxblox pieces that are
 * not vendored: MeshIntegrator::generateMesh (marching cubes over the dual cells of the
 * TSDF, a cell is meshed iff all 8 corner voxels have weight > min_weight),
 * MeshLayer::getConnectedMesh(mesh, 0.5 * voxel_size) (vertices whose coordinates round to
 * the same 0.5-voxel cell collapse onto the FIRST one met while iterating an unordered_map
 * of blocks -- implementation-defined order) and Interpolator::getVoxel(..., interpolate =
 * true).  Restated [recalled] as a vertex SET: one vertex per sign-changing cell edge that
 * belongs to at least one fully observed cell (every marching-cubes configuration uses all
 * of its sign-changing edges), interpolated low-to-high along the edge; duplicates are
 * resolved canonically (block order, voxel linear index, axis) because the reference's
 * own choice is unspecified.
 */
#ifndef VOXGRAPH_AMD_ORACLE_ISO_ORACLE_H_
#define VOXGRAPH_AMD_ORACLE_ISO_ORACLE_H_
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif
/* Layers in voxblox layout.  Outputs (nullable to count only): xyz[n][3], distance[n],
 * weight[n] (interpolated TSDF distance / weight, voxgraph_submap.cpp:226-235).
 * Returns the number of isosurface points or -1. */
int64_t orc_isosurface_points(float voxel_size, int vps, int n_blocks,
                              const int32_t* block_index, const float* tsdf_distance,
                              const float* tsdf_weight, float min_weight, float* xyz,
                              float* distance, float* weight);
#ifdef __cplusplus
}
#endif
#endif
