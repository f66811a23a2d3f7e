"""ctypes view of oracle/_ref/libref_reg.so: the REFERENCE's own
voxgraph::RegistrationCostFunction, voxgraph::VoxgraphSubmap and voxgraph::BoundingBox
(registration_cost_function.cpp, voxgraph_submap.cpp, bounding_box.cpp compiled from
/root/reference against the stand-in headers of oracle/ref_shims; see its README).

TEST INFRASTRUCTURE.  Only tests/ and tests/golden/make_ref_golden.py import this.
`available()` is False wherever the library was not built (it can only be built where
/root/reference exists; the built .so travels to the GPU box with the snapshot)."""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
# VGX_REF_DIR: another build of the same driver (oracle/PIN.md: the reference sources against the REAL voxblox headers)
_SO = os.path.join(os.environ.get("VGX_REF_DIR") or os.path.join(_HERE, "_ref"), "libref_reg.so")
_LIB = None

POINTS_ISOSURFACE, POINTS_VOXELS = 0, 1


def build():
    """(Re)build when the reference sources are present; no-op otherwise."""
    subprocess.run(["make", "-s", "-C", _HERE, "ref"], check=True)
    return os.path.exists(_SO)


def available():
    return os.path.exists(_SO)


def _lib():
    global _LIB
    if _LIB is None:
        lib = C.CDLL(_SO)
        vp, f32p, f64p = C.c_void_p, C.POINTER(C.c_float), C.POINTER(C.c_double)
        i32p = C.POINTER(C.c_int32)
        lib.refreg_submap_create.restype = vp
        lib.refreg_submap_create.argtypes = [C.c_uint32, f64p, C.c_float, C.c_int32, C.c_int32,
                                             i32p, f32p, f32p, f32p, C.POINTER(C.c_uint8),
                                             C.c_double, C.c_double, C.c_int32]
        lib.refreg_submap_destroy.argtypes = [vp]
        lib.refreg_submap_set_pose.argtypes = [vp, f64p]
        lib.refreg_submap_block_order.restype = C.c_int32
        lib.refreg_submap_block_order.argtypes = [vp, i32p]
        lib.refreg_submap_num_points.restype = C.c_int64
        lib.refreg_submap_num_points.argtypes = [vp, C.c_int32]
        lib.refreg_submap_get_points.argtypes = [vp, C.c_int32, f32p, f32p, f32p]
        lib.refreg_submap_set_points.argtypes = [vp, C.c_int32, C.c_int64, f32p, f32p, f32p]
        lib.refreg_submap_isosurface_blocks.restype = C.c_int32
        lib.refreg_submap_isosurface_blocks.argtypes = [vp, i32p]
        lib.refreg_submap_surface_obb.argtypes = [vp, f32p]
        lib.refreg_submap_mission_surface_aabb.argtypes = [vp, f32p]
        lib.refreg_submap_overlaps_with.restype = C.c_int32
        lib.refreg_submap_overlaps_with.argtypes = [vp, vp]
        lib.refreg_cost_create.restype = vp
        lib.refreg_cost_create.argtypes = [vp, vp, C.c_int32, C.c_float, C.c_double, C.c_int32]
        lib.refreg_cost_destroy.argtypes = [vp]
        lib.refreg_cost_num_residuals.restype = C.c_int32
        lib.refreg_cost_num_residuals.argtypes = [vp]
        lib.refreg_cost_evaluate.restype = C.c_int32
        lib.refreg_cost_evaluate.argtypes = [vp, f64p, f64p, C.c_int32, f64p, f64p, f64p]
        lib.refreg_relative_pose_residual.restype = C.c_int32
        lib.refreg_relative_pose_residual.argtypes = [f64p, f64p, f64p, f64p, f64p, f64p]
        _LIB = lib
    return _LIB


def _ptr(a, t):
    return None if a is None else a.ctypes.data_as(C.POINTER(t))


class Submap:
    """The reference's VoxgraphSubmap, constructed from a TSDF layer, ESDF voxels injected,
    then finishSubmap() (voxgraph_submap.cpp:84-107): surface OBB, findRelevantVoxelIndices,
    findIsosurfaceVertices with the given registration filter (voxgraph_submap.h:26-30)."""

    def __init__(self, submap_id, pose, voxel_size, vps, block_index, tsdf_distance, tsdf_weight,
                 esdf_distance=None, esdf_observed=None, min_voxel_weight=1.0, max_voxel_distance=0.3,
                 use_esdf_distance=True):
        bi = np.ascontiguousarray(block_index, np.int32).reshape(-1, 3)
        td = np.ascontiguousarray(tsdf_distance, np.float32).ravel()
        tw = np.ascontiguousarray(tsdf_weight, np.float32).ravel()
        ed = None if esdf_distance is None else np.ascontiguousarray(esdf_distance, np.float32).ravel()
        eo = None if esdf_observed is None else np.ascontiguousarray(esdf_observed, np.uint8).ravel()
        pose = np.ascontiguousarray(pose, np.float64)
        self._h = _lib().refreg_submap_create(int(submap_id), _ptr(pose, C.c_double), float(voxel_size),
                                              int(vps), bi.shape[0], _ptr(bi, C.c_int32),
                                              _ptr(td, C.c_float), _ptr(tw, C.c_float),
                                              _ptr(ed, C.c_float), _ptr(eo, C.c_uint8),
                                              float(min_voxel_weight), float(max_voxel_distance),
                                              int(bool(use_esdf_distance)))

    def set_pose(self, pose):
        pose = np.ascontiguousarray(pose, np.float64)
        _lib().refreg_submap_set_pose(self._h, _ptr(pose, C.c_double))

    def block_order(self):
        """Layer::getAllAllocatedBlocks order (hash-map order: what the reference iterates in)."""
        n = _lib().refreg_submap_block_order(self._h, None)
        out = np.zeros((n, 3), np.int32)
        _lib().refreg_submap_block_order(self._h, _ptr(out, C.c_int32))
        return out

    def points(self, point_type):
        n = int(_lib().refreg_submap_num_points(self._h, int(point_type)))
        xyz, d, w = np.zeros((n, 3), np.float32), np.zeros(n, np.float32), np.zeros(n, np.float32)
        if n:
            _lib().refreg_submap_get_points(self._h, int(point_type), _ptr(xyz, C.c_float),
                                            _ptr(d, C.c_float), _ptr(w, C.c_float))
        return xyz, d, w

    def set_points(self, point_type, xyz, distance, weight):
        """Replace a sampler's contents (addItem(point, weight) per point, as the reference fills it)."""
        xyz = np.ascontiguousarray(xyz, np.float32).reshape(-1, 3)
        d = np.ascontiguousarray(distance, np.float32)
        w = np.ascontiguousarray(weight, np.float32)
        _lib().refreg_submap_set_points(self._h, int(point_type), xyz.shape[0], _ptr(xyz, C.c_float),
                                        _ptr(d, C.c_float), _ptr(w, C.c_float))

    def isosurface_blocks(self):
        n = _lib().refreg_submap_isosurface_blocks(self._h, None)
        out = np.zeros((n, 3), np.int32)
        if n:
            _lib().refreg_submap_isosurface_blocks(self._h, _ptr(out, C.c_int32))
        return out

    def surface_obb(self):
        out = np.zeros(6, np.float32)
        _lib().refreg_submap_surface_obb(self._h, _ptr(out, C.c_float))
        return out[:3].copy(), out[3:].copy()

    def mission_surface_aabb(self):
        out = np.zeros(6, np.float32)
        _lib().refreg_submap_mission_surface_aabb(self._h, _ptr(out, C.c_float))
        return out[:3].copy(), out[3:].copy()

    def overlapsWith(self, other):
        return bool(_lib().refreg_submap_overlaps_with(self._h, other._h))

    def __del__(self):
        if getattr(self, "_h", None):
            _lib().refreg_submap_destroy(self._h)
            self._h = None


class RegistrationCostFunction:
    """voxgraph::RegistrationCostFunction(reference_submap, reading_submap, config)."""

    def __init__(self, reference, reading, point_type=POINTS_VOXELS, sampling_ratio=-1.0,
                 no_correspondence_cost=0.0, use_esdf_distance=True):
        self._keep = (reference, reading)
        self._h = _lib().refreg_cost_create(reference._h, reading._h, int(point_type),
                                            float(sampling_ratio), float(no_correspondence_cost),
                                            int(bool(use_esdf_distance)))

    def num_residuals(self):
        return int(_lib().refreg_cost_num_residuals(self._h))

    def Evaluate(self, ref_pose, read_pose, want_jac=True, want_ref=True, want_read=True):
        n = self.num_residuals()
        r = np.zeros(n, np.float64)
        j0 = np.zeros((n, 4), np.float64) if (want_jac and want_ref) else None
        j1 = np.zeros((n, 4), np.float64) if (want_jac and want_read) else None
        a = np.ascontiguousarray(ref_pose, np.float64)
        b = np.ascontiguousarray(read_pose, np.float64)
        ok = _lib().refreg_cost_evaluate(self._h, _ptr(a, C.c_double), _ptr(b, C.c_double),
                                         1 if want_jac else 0, _ptr(r, C.c_double),
                                         _ptr(j0, C.c_double), _ptr(j1, C.c_double))
        return bool(ok), r, j0, j1

    def __del__(self):
        if getattr(self, "_h", None):
            _lib().refreg_cost_destroy(self._h)
            self._h = None


def relative_pose_residual(observed_xyz_yaw, sqrt_information, pose_a, pose_b):
    """voxgraph::RelativePoseCostFunction (T = double) -> (residuals[4], observed values as stored)."""
    obs = np.ascontiguousarray(observed_xyz_yaw, np.float64)
    info = np.ascontiguousarray(sqrt_information, np.float64).reshape(4, 4)
    a, b = np.ascontiguousarray(pose_a, np.float64), np.ascontiguousarray(pose_b, np.float64)
    r, stored = np.zeros(4), np.zeros(4)
    ok = _lib().refreg_relative_pose_residual(_ptr(obs, C.c_double), _ptr(info, C.c_double), _ptr(a, C.c_double),
                                              _ptr(b, C.c_double), _ptr(r, C.c_double), _ptr(stored, C.c_double))
    assert ok
    return r, stored
