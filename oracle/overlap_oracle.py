"""CPU restatement (numpy, small cases) of voxgraph's overlap test.  TEST INFRASTRUCTURE.

  VoxgraphSubmap::getSubmapFrameSurfaceObb   voxgraph/src/frontend/submap_collection/voxgraph_submap.cpp:280-321
  BoundingBox::getAabbFromObbAndPose         voxgraph/src/frontend/submap_collection/bounding_box.cpp:28-42
  VoxgraphSubmap::overlapsWith               voxgraph_submap.cpp:245-278
  PoseGraphInterface::updateOverlappingSubmapList  voxgraph/src/frontend/pose_graph_interface/pose_graph_interface.cpp:109-147

This part of the path is in the reference itself (no un-vendored arithmetic except the f32 rigid
transform, taken from oracle/reg_oracle.c).  Pinned: tests/test_ref_submap_pin.py runs the
reference's own voxgraph_submap.cpp / bounding_box.cpp (oracle/_ref, compiled against
oracle/ref_shims) and these functions reproduce its boxes and overlap decisions exactly.
"""
import numpy as np

from . import pyoracle as orc

F = np.float32


def surface_obb(voxel_points_xyz, voxel_size):
    half = F(0.5) * F(voxel_size)
    return voxel_points_xyz.min(0).astype(F) - half, voxel_points_xyz.max(0).astype(F) + half


def mission_aabb(obb_min, obb_max, pose):
    q, t = orc.relative_transform(pose, np.zeros(4))          # exp(0)^-1 * exp(pose)
    pts = []
    for i in range(8):
        c = [obb_min[a] if (i >> a) & 1 else obb_max[a] for a in range(3)]
        pts.append(orc.transform_point(q, t, np.array(c, F)))
    pts = np.array(pts, F)
    return pts.min(0), pts.max(0)


def isosurface_blocks(iso_xyz, voxel_size, vps):
    bsi = F(1.0) / (F(vps) * F(voxel_size))
    b = np.floor(iso_xyz.astype(F) * bsi + F(1e-6)).astype(np.int64)
    return np.unique(b, axis=0)


def overlaps_with(sub_a, pose_a, sub_b, pose_b):
    """sub = dict(voxel_size, vps, block_index, voxel_xyz, iso_xyz)"""
    amn, amx = mission_aabb(*surface_obb(sub_a["voxel_xyz"], sub_a["voxel_size"]), pose_a)
    bmn, bmx = mission_aabb(*surface_obb(sub_b["voxel_xyz"], sub_b["voxel_size"]), pose_b)
    for a in range(3):
        if amx[a] < bmn[a] or amn[a] > bmx[a]:
            return False
    q, t = orc.relative_transform(pose_a, pose_b)             # other^-1 * current
    bs_a = F(sub_a["vps"]) * F(sub_a["voxel_size"])
    bsi_b = F(1.0) / (F(sub_b["vps"]) * F(sub_b["voxel_size"]))
    blocks_b = set(map(tuple, np.asarray(sub_b["block_index"]).astype(np.int64)))
    for blk in isosurface_blocks(sub_a["iso_xyz"], sub_a["voxel_size"], sub_a["vps"]):
        centre = ((blk.astype(F) + F(0.5)) * bs_a).astype(F)
        p = orc.transform_point(q, t, centre)
        other = tuple(int(v) for v in np.floor(p * bsi_b + F(1e-6)))
        if other in blocks_b:
            return True
    return False


def overlapping_pairs(subs, poses):
    return [(i, j) for i in range(len(subs)) for j in range(i + 1, len(subs))
            if overlaps_with(subs[i], poses[i], subs[j], poses[j])]
