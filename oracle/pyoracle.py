"""ctypes bindings for the CPU oracle (oracle/_build/liboracle.so).

TEST INFRASTRUCTURE ONLY -- see oracle/reg_oracle.h.  Only tests/,
__graft_entry__.smoke() and bench.py's cpu_baseline leg import this module.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "_build", "liboracle.so")


_NATIVE = False


def use_native_build():
    """bench.py's cpu_baseline: BASELINE.md promises the CPU port built `-O3 -march=native`.  Such a
    library only runs on the host it was built on, so it is (re)built HERE, now, into
    _build/liboracle_native.so (oracle/Makefile `native`: same sources, same -ffp-contract=off
    -fno-fast-math, so the same IEEE results) and loaded instead of the portable -O2 build the tests
    use.  Must be called before the first use of the library; returns False (and keeps the portable
    build) if the compile fails."""
    global _LIB_PATH, _NATIVE
    if _lib is not None:
        return _NATIVE
    try:
        subprocess.check_call(["make", "-C", _HERE, "-s", "-B", "native"])
        path = os.path.join(_HERE, "_build", "liboracle_native.so")
        C.CDLL(path)
        _LIB_PATH, _NATIVE = path, True
    except Exception:
        _NATIVE = False
    return _NATIVE


def build_flags():
    return "gcc -O3 -march=native, built on this host" if _NATIVE else "gcc -O2, portable build"


def build(force=False):
    """Compile the oracle with gcc (oracle/Makefile)."""
    if _NATIVE:
        return _LIB_PATH
    if force or not os.path.exists(_LIB_PATH) or _stale():
        subprocess.check_call(["make", "-C", _HERE, "-s"])
    return _LIB_PATH


def _stale():
    t = os.path.getmtime(_LIB_PATH)
    for f in os.listdir(_HERE):
        if f.endswith(("_oracle.c", "_oracle.h")):
            if os.path.getmtime(os.path.join(_HERE, f)) > t:
                return True
    return False


_lib = None

c_f32p = C.POINTER(C.c_float)
c_f64p = C.POINTER(C.c_double)
c_i32p = C.POINTER(C.c_int32)
c_i64p = C.POINTER(C.c_int64)
c_u8p = C.POINTER(C.c_uint8)


class _OrcLayer(C.Structure):
    _fields_ = [
        ("voxel_size", C.c_float), ("voxel_size_inv", C.c_float),
        ("block_size", C.c_float), ("block_size_inv", C.c_float),
        ("vps", C.c_int), ("n_blocks", C.c_int),
        ("block_index", c_i32p), ("distance", c_f32p), ("valid", c_u8p),
        ("lut_min", C.c_int32 * 3), ("lut_dim", C.c_int32 * 3),
        ("lut", c_i32p),
    ]


class _OrcRegConfig(C.Structure):
    _fields_ = [("no_correspondence_cost", C.c_double)]


class _Mt(C.Structure):
    _fields_ = [("mt", C.c_uint32 * 624), ("idx", C.c_int)]


def lib():
    global _lib
    if _lib is None:
        build()
        L = C.CDLL(_LIB_PATH)
        L.orc_layer_init.argtypes = [C.POINTER(_OrcLayer), C.c_float, C.c_int, C.c_int,
                                     c_i32p, c_f32p, c_u8p]
        L.orc_layer_init.restype = C.c_int
        L.orc_layer_free.argtypes = [C.POINTER(_OrcLayer)]
        L.orc_get_voxels_and_q.argtypes = [C.POINTER(_OrcLayer), c_f32p, c_f32p, c_f32p]
        L.orc_get_voxels_and_q.restype = C.c_int
        L.orc_relative_transform.argtypes = [c_f64p, c_f64p, c_f32p, c_f32p]
        L.orc_transform_point.argtypes = [c_f32p, c_f32p, c_f32p, c_f32p]
        L.orc_pose_jacobian_matrices.argtypes = [C.c_float, C.c_float, c_f64p, c_f64p,
                                                 c_f32p, c_f32p]
        L.orc_reg_evaluate.argtypes = [C.POINTER(_OrcLayer), C.POINTER(_OrcRegConfig),
                                       C.c_int64, c_f32p, c_f32p, c_f32p, c_i64p,
                                       c_f64p, c_f64p, C.c_int, c_f64p, c_f64p, c_f64p]
        L.orc_reg_evaluate.restype = C.c_int
        L.orc_reg_evaluate_normal.argtypes = [C.POINTER(_OrcLayer), C.POINTER(_OrcRegConfig),
                                              C.c_int64, c_f32p, c_f32p, c_f32p,
                                              c_f64p, c_f64p, c_f64p, c_f64p, c_f64p]
        L.orc_reg_evaluate_normal.restype = C.c_int
        L.orc_mt19937_seed.argtypes = [C.POINTER(_Mt), C.c_uint32]
        L.orc_mt19937_next.argtypes = [C.POINTER(_Mt)]
        L.orc_mt19937_next.restype = C.c_uint32
        L.orc_uniform01.argtypes = [C.POINTER(_Mt)]
        L.orc_uniform01.restype = C.c_double
        L.orc_weighted_draw.argtypes = [C.POINTER(_Mt), c_f64p, C.c_int64]
        L.orc_weighted_draw.restype = C.c_int64
        L.orc_find_relevant_voxels.argtypes = [C.c_float, C.c_int, C.c_int, c_i32p, c_f32p,
                                               c_f32p, c_f32p, C.c_double, C.c_double,
                                               c_f32p, c_f32p, c_f32p]
        L.orc_find_relevant_voxels.restype = C.c_int64
        _lib = L
    return _lib


def _p(a, t):
    return None if a is None else a.ctypes.data_as(t)


def _f32(a):
    return np.ascontiguousarray(a, dtype=np.float32)


def _f64(a):
    return np.ascontiguousarray(a, dtype=np.float64)


class Layer:
    """voxblox::Layer stand-in: distance + validity per voxel, block-sparse."""

    def __init__(self, voxel_size, vps, block_index, distance, valid):
        self.block_index = np.ascontiguousarray(block_index, dtype=np.int32).reshape(-1, 3)
        n = self.block_index.shape[0]
        self.distance = _f32(distance).reshape(n, vps ** 3)
        self.valid = np.ascontiguousarray(valid, dtype=np.uint8).reshape(n, vps ** 3)
        self.voxel_size = float(np.float32(voxel_size))
        self.vps = int(vps)
        self._c = _OrcLayer()
        rc = lib().orc_layer_init(C.byref(self._c), self.voxel_size, self.vps, n,
                                  _p(self.block_index, c_i32p), _p(self.distance, c_f32p),
                                  _p(self.valid, c_u8p))
        if rc != 0:
            raise MemoryError("orc_layer_init failed")

    def __del__(self):
        try:
            lib().orc_layer_free(C.byref(self._c))
        except Exception:
            pass

    def voxels_and_q(self, pos):
        pos = _f32(pos)
        d = np.zeros(8, np.float32)
        q = np.zeros(8, np.float32)
        ok = lib().orc_get_voxels_and_q(C.byref(self._c), _p(pos, c_f32p), _p(d, c_f32p),
                                        _p(q, c_f32p))
        return bool(ok), d, q


def relative_transform(ref_pose, read_pose):
    q = np.zeros(4, np.float32)
    t = np.zeros(3, np.float32)
    lib().orc_relative_transform(_p(_f64(ref_pose), c_f64p), _p(_f64(read_pose), c_f64p),
                                 _p(q, c_f32p), _p(t, c_f32p))
    return q, t


def transform_point(q, t, p):
    out = np.zeros(3, np.float32)
    lib().orc_transform_point(_p(_f32(q), c_f32p), _p(_f32(t), c_f32p), _p(_f32(p), c_f32p),
                              _p(out, c_f32p))
    return out


def pose_jacobian_matrices(xi, yi, ref_pose, read_pose):
    mo = np.zeros(12, np.float32)
    me = np.zeros(12, np.float32)
    lib().orc_pose_jacobian_matrices(float(xi), float(yi), _p(_f64(ref_pose), c_f64p),
                                     _p(_f64(read_pose), c_f64p), _p(mo, c_f32p),
                                     _p(me, c_f32p))
    return mo.reshape(3, 4), me.reshape(3, 4)


def reg_evaluate(reading, xyz, dist, weight, ref_pose, read_pose, want_jac=True,
                 want_ref=True, want_read=True, no_correspondence_cost=0.0,
                 sample_idx=None):
    """RegistrationCostFunction::Evaluate.  Returns (ok, residuals, jac_ref, jac_read)."""
    xyz = _f32(xyz).reshape(-1, 3)
    dist = _f32(dist)
    weight = _f32(weight)
    if sample_idx is not None:
        sample_idx = np.ascontiguousarray(sample_idx, dtype=np.int64)
        n = sample_idx.shape[0]
    else:
        n = xyz.shape[0]
    res = np.zeros(n, np.float64)
    jr = np.zeros((n, 4), np.float64) if (want_jac and want_ref) else None
    je = np.zeros((n, 4), np.float64) if (want_jac and want_read) else None
    cfg = _OrcRegConfig(no_correspondence_cost)
    ok = lib().orc_reg_evaluate(C.byref(reading._c), C.byref(cfg), n, _p(xyz, c_f32p),
                                _p(dist, c_f32p), _p(weight, c_f32p), _p(sample_idx, c_i64p),
                                _p(_f64(ref_pose), c_f64p), _p(_f64(read_pose), c_f64p),
                                1 if want_jac else 0, _p(res, c_f64p), _p(jr, c_f64p),
                                _p(je, c_f64p))
    return bool(ok), res, jr, je


def reg_evaluate_normal(reading, xyz, dist, weight, ref_pose, read_pose,
                        no_correspondence_cost=0.0):
    xyz = _f32(xyz).reshape(-1, 3)
    dist = _f32(dist)
    weight = _f32(weight)
    cost = np.zeros(1, np.float64)
    jtr = np.zeros(8, np.float64)
    jtj = np.zeros(36, np.float64)
    cfg = _OrcRegConfig(no_correspondence_cost)
    ok = lib().orc_reg_evaluate_normal(C.byref(reading._c), C.byref(cfg), xyz.shape[0],
                                       _p(xyz, c_f32p), _p(dist, c_f32p), _p(weight, c_f32p),
                                       _p(_f64(ref_pose), c_f64p), _p(_f64(read_pose), c_f64p),
                                       _p(cost, c_f64p), _p(jtr, c_f64p), _p(jtj, c_f64p))
    return bool(ok), float(cost[0]), jtr, jtj


class Mt19937:
    def __init__(self, seed=5489):
        self._g = _Mt()
        lib().orc_mt19937_seed(C.byref(self._g), seed)

    def next(self):
        return int(lib().orc_mt19937_next(C.byref(self._g)))

    def uniform01(self):
        return float(lib().orc_uniform01(C.byref(self._g)))

    def weighted_draw(self, cumulative):
        cumulative = _f64(cumulative)
        return int(lib().orc_weighted_draw(C.byref(self._g), _p(cumulative, c_f64p),
                                           cumulative.shape[0]))


def find_relevant_voxels(voxel_size, vps, block_index, tsdf_distance, tsdf_weight,
                         esdf_distance, min_voxel_weight=1.0, max_voxel_distance=0.3):
    """VoxgraphSubmap::findRelevantVoxelIndices -> (xyz[n,3], dist[n], weight[n])."""
    bi = np.ascontiguousarray(block_index, dtype=np.int32).reshape(-1, 3)
    td = _f32(tsdf_distance)
    tw = _f32(tsdf_weight)
    ed = None if esdf_distance is None else _f32(esdf_distance)
    args = (float(np.float32(voxel_size)), int(vps), bi.shape[0], _p(bi, c_i32p),
            _p(td, c_f32p), _p(tw, c_f32p), _p(ed, c_f32p), float(min_voxel_weight),
            float(max_voxel_distance))
    n = lib().orc_find_relevant_voxels(*args, None, None, None)
    xyz = np.zeros((n, 3), np.float32)
    dist = np.zeros(n, np.float32)
    weight = np.zeros(n, np.float32)
    lib().orc_find_relevant_voxels(*args, _p(xyz, c_f32p), _p(dist, c_f32p),
                                   _p(weight, c_f32p))
    return xyz, dist, weight


# ----------------------------------------------------------------------------
# TSDF path (oracle/tsdf_oracle.c)
# ----------------------------------------------------------------------------
class TsdfConfig(C.Structure):
    """voxblox::TsdfIntegratorBase::Config [recalled]; same field order as orc_tsdf_config
    and vgx_tsdf_config."""
    _fields_ = [("default_truncation_distance", C.c_float), ("max_weight", C.c_float),
                ("voxel_carving_enabled", C.c_int), ("min_ray_length_m", C.c_float),
                ("max_ray_length_m", C.c_float), ("use_const_weight", C.c_int),
                ("allow_clear", C.c_int), ("use_weight_dropoff", C.c_int),
                ("use_sparsity_compensation_factor", C.c_int),
                ("sparsity_compensation_factor", C.c_float),
                ("start_voxel_subsampling_factor", C.c_float),
                ("max_consecutive_ray_collisions", C.c_int),
                ("clear_checks_every_n_frames", C.c_int), ("integration_order", C.c_int),
                ("enable_anti_grazing", C.c_int)]


class ReplayLayer(C.Structure):
    """orc_replay_layer (oracle/tsdf_replay.h): the arrays of a layer download"""
    _fields_ = [("n_blocks", C.c_int32), ("block_index", c_i32p), ("distance", c_f32p), ("weight", c_f32p), ("rgba", c_u8p)]


class ReplayReport(C.Structure):
    """orc_replay_report (oracle/tsdf_replay.h)"""
    _names = ("points", "valid_points", "start_exchanges", "start_skips", "rays_cast", "rays_bad", "observed_exchanges",
              "overrun_exchanges", "rays_with_overrun", "max_overrun", "rays_stopped_early", "rays_walked_to_end",
              "required_updates", "fold_events", "folds_published", "folds_left_alone", "fold_records", "longest_fold",
              "colour_writes", "start_slots_touched", "observed_slots_touched", "voxels_touched", "voxels_with_several_links",
              "new_blocks", "errors")
    _fields_ = [(k, C.c_int64) for k in _names] + [("first_error", C.c_char * 400)]

    def as_dict(self):
        d = {k: int(getattr(self, k)) for k in self._names}
        d["first_error"] = self.first_error.decode(errors="replace")
        return d


_tsdf_bound = False


def _tsdf_lib():
    global _tsdf_bound
    L = lib()
    if not _tsdf_bound:
        L.orc_tsdf_config_default.argtypes = [C.POINTER(TsdfConfig)]
        L.orc_tsdf_layer_create.argtypes = [C.c_float, C.c_int]
        L.orc_tsdf_layer_create.restype = C.c_void_p
        L.orc_tsdf_layer_destroy.argtypes = [C.c_void_p]
        L.orc_tsdf_layer_num_blocks.argtypes = [C.c_void_p]
        L.orc_tsdf_layer_num_blocks.restype = C.c_int
        L.orc_tsdf_layer_download.argtypes = [C.c_void_p, c_i32p, c_f32p, c_f32p, c_u8p]
        L.orc_tsdf_integrator_create.argtypes = [C.POINTER(TsdfConfig), C.c_void_p]
        L.orc_tsdf_integrator_create.restype = C.c_void_p
        L.orc_tsdf_integrator_destroy.argtypes = [C.c_void_p]
        L.orc_tsdf_integrator_set_layer.argtypes = [C.c_void_p, C.c_void_p]
        L.orc_tsdf_integrate.argtypes = [C.c_void_p, c_f32p, c_f32p, c_u8p, C.c_int64, C.c_int]
        L.orc_tsdf_integrate.restype = C.c_int64
        L.orc_tsdf_merged_integrate.argtypes = [C.c_void_p, c_f32p, c_f32p, c_u8p, C.c_int64, C.c_int]
        L.orc_tsdf_merged_integrate.restype = C.c_int64
        L.orc_tsdf_integrate_sequence.argtypes = [C.c_void_p, C.c_int, c_f32p, c_f32p, C.c_int64, C.c_int]
        L.orc_tsdf_integrate_sequence.restype = C.c_int64
        c_u64p = C.POINTER(C.c_uint64)
        L.orc_tsdf_integrator_set_log.argtypes = [C.c_void_p, c_u64p, C.c_int64]
        L.orc_tsdf_integrator_set_log.restype = None
        L.orc_tsdf_integrator_log_words.argtypes = [C.c_void_p, C.POINTER(C.c_int64)]
        L.orc_tsdf_integrator_log_words.restype = C.c_int64
        L.orc_tsdf_integrator_download_sets.argtypes = [C.c_void_p, c_u64p, c_u64p, c_u64p]
        L.orc_tsdf_integrator_download_sets.restype = None
        L.orc_tsdf_replay_check.argtypes = [C.POINTER(TsdfConfig), C.c_float, C.c_int, c_f32p, c_f32p, c_u8p, C.c_int64, C.c_int,
                                            C.c_uint64, C.c_uint64, c_u64p, c_u64p, c_u64p, c_u64p,
                                            C.POINTER(ReplayLayer), C.POINTER(ReplayLayer), c_u64p, C.c_int64,
                                            C.POINTER(ReplayReport)]
        L.orc_tsdf_replay_check.restype = C.c_int64
        _tsdf_bound = True
    return L


def tsdf_config(**kw):
    cfg = TsdfConfig()
    _tsdf_lib().orc_tsdf_config_default(C.byref(cfg))
    for k, v in kw.items():
        if not hasattr(cfg, k):
            raise AttributeError(k)
        setattr(cfg, k, v)
    return cfg


def voxgraph_tsdf_config(**kw):
    """voxgraph/config/voxgraph_mapper.yaml:21-28 over voxblox's defaults."""
    base = dict(default_truncation_distance=0.60, max_ray_length_m=16.0, use_const_weight=1,
                use_weight_dropoff=1, use_sparsity_compensation_factor=1,
                sparsity_compensation_factor=20.0)
    base.update(kw)
    return tsdf_config(**base)


class TsdfLayer:
    def __init__(self, voxel_size, vps=16):
        self.voxel_size, self.vps = float(np.float32(voxel_size)), vps
        self.h = _tsdf_lib().orc_tsdf_layer_create(self.voxel_size, vps)

    def num_blocks(self):
        return _tsdf_lib().orc_tsdf_layer_num_blocks(self.h)

    def download(self):
        n, nv = self.num_blocks(), self.vps ** 3
        bi = np.zeros((n, 3), np.int32)
        d = np.zeros((n, nv), np.float32)
        w = np.zeros((n, nv), np.float32)
        rgba = np.zeros((n, nv, 4), np.uint8)
        _tsdf_lib().orc_tsdf_layer_download(self.h, _p(bi, c_i32p), _p(d, c_f32p), _p(w, c_f32p),
                                            _p(rgba, c_u8p))
        return bi, d, w, rgba

    def __del__(self):
        try:
            _tsdf_lib().orc_tsdf_layer_destroy(self.h)
        except Exception:
            pass


class FastTsdfIntegrator:
    """voxblox::FastTsdfIntegrator(config, layer) / setLayer / integratePointCloud."""

    def __init__(self, config, layer):
        self.config, self.layer = config, layer
        self.h = _tsdf_lib().orc_tsdf_integrator_create(C.byref(config), layer.h)

    def setLayer(self, layer):
        self.layer = layer
        _tsdf_lib().orc_tsdf_integrator_set_layer(self.h, layer.h)

    def integratePointCloud(self, T_G_C, points_C, colors=None, freespace_points=False):
        T = _f32(T_G_C)
        pts = _f32(points_C).reshape(-1, 3)
        col = None if colors is None else np.ascontiguousarray(colors, np.uint8).reshape(-1, 4)
        return _tsdf_lib().orc_tsdf_integrate(self.h, _p(T, c_f32p), _p(pts, c_f32p),
                                              _p(col, c_u8p), pts.shape[0], int(freespace_points))

    def set_log(self, capacity_words):
        """scans from now on append their events (the racing kernel's log format) to a buffer of that many words; 0: off"""
        self._log = np.zeros(int(capacity_words), np.uint64) if capacity_words else None
        _tsdf_lib().orc_tsdf_integrator_set_log(self.h, _p(self._log, C.POINTER(C.c_uint64)), int(capacity_words))

    def read_log(self):
        """-> (the words logged since set_log / the last read_log, events lost); empties the log"""
        lost = C.c_int64()
        n = _tsdf_lib().orc_tsdf_integrator_log_words(self.h, C.byref(lost))
        out = self._log[:n].copy()
        _tsdf_lib().orc_tsdf_integrator_set_log(self.h, _p(self._log, C.POINTER(C.c_uint64)), len(self._log))
        return out, int(lost.value)

    def download_sets(self):
        """-> (start set, observed set: uint64[2^20] each, (start offset, observed offset))"""
        a, b, o = np.zeros(1 << 20, np.uint64), np.zeros(1 << 20, np.uint64), np.zeros(2, np.uint64)
        u = C.POINTER(C.c_uint64)
        _tsdf_lib().orc_tsdf_integrator_download_sets(self.h, _p(a, u), _p(b, u), _p(o, u))
        return a, b, (int(o[0]), int(o[1]))

    def integrate_sequence(self, poses, clouds, repeats=1):
        """`repeats` passes over the scans (poses [k][7], clouds [k][n][3]) inside ONE foreign call (no
        interpreter between scans: bench.py's all-cores timing); returns the number of voxel updates"""
        P = _f32(poses).reshape(-1, 7)
        pts = _f32(clouds).reshape(P.shape[0], -1, 3)
        return _tsdf_lib().orc_tsdf_integrate_sequence(self.h, P.shape[0], _p(P, c_f32p), _p(pts, c_f32p),
                                                       pts.shape[1], int(repeats))

    def integratePointCloudMerged(self, T_G_C, points_C, colors=None, freespace_points=False):
        """voxblox::MergedTsdfIntegrator::integratePointCloud on the same config / layer"""
        T = _f32(T_G_C)
        pts = _f32(points_C).reshape(-1, 3)
        col = None if colors is None else np.ascontiguousarray(colors, np.uint8).reshape(-1, 4)
        return _tsdf_lib().orc_tsdf_merged_integrate(self.h, _p(T, c_f32p), _p(pts, c_f32p),
                                                     _p(col, c_u8p), pts.shape[0], int(freespace_points))

    def __del__(self):
        try:
            _tsdf_lib().orc_tsdf_integrator_destroy(self.h)
        except Exception:
            pass


# ----------------------------------------------------------------------------
# ESDF generation (oracle/esdf_oracle.c)
# ----------------------------------------------------------------------------
class EsdfConfig(C.Structure):
    """voxblox::EsdfIntegrator::Config defaults [recalled]; same layout as vgx_esdf_config."""
    _fields_ = [("max_distance_m", C.c_float), ("min_distance_m", C.c_float),
                ("default_distance_m", C.c_float), ("min_diff_m", C.c_float),
                ("min_weight", C.c_float), ("num_buckets", C.c_int)]


def esdf_config(**kw):
    cfg = EsdfConfig()
    L = lib()
    L.orc_esdf_config_default.argtypes = [C.POINTER(EsdfConfig)]
    L.orc_esdf_config_default(C.byref(cfg))
    for k, v in kw.items():
        if not hasattr(cfg, k):
            raise AttributeError(k)
        setattr(cfg, k, v)
    return cfg


def esdf_from_tsdf(voxel_size, vps, block_index, tsdf_distance, tsdf_weight, config=None):
    """EsdfIntegrator::updateFromTsdfLayerBatch -> (esdf_distance, esdf_observed, n_updates)."""
    cfg = config or esdf_config()
    L = lib()
    L.orc_esdf_from_tsdf_batch.argtypes = [C.POINTER(EsdfConfig), C.c_float, C.c_int, C.c_int,
                                           c_i32p, c_f32p, c_f32p, c_f32p, c_u8p]
    L.orc_esdf_from_tsdf_batch.restype = C.c_int64
    bi = np.ascontiguousarray(block_index, np.int32).reshape(-1, 3)
    td, tw = _f32(tsdf_distance), _f32(tsdf_weight)
    ed = np.zeros_like(td)
    eo = np.zeros(td.shape, np.uint8)
    n = L.orc_esdf_from_tsdf_batch(C.byref(cfg), float(np.float32(voxel_size)), vps, bi.shape[0],
                                   _p(bi, c_i32p), _p(td, c_f32p), _p(tw, c_f32p), _p(ed, c_f32p),
                                   _p(eo, c_u8p))
    if n < 0:
        raise MemoryError("orc_esdf_from_tsdf_batch")
    return ed, eo, n


# ----------------------------------------------------------------------------
# isosurface registration points (oracle/iso_oracle.c)
# ----------------------------------------------------------------------------
def isosurface_points(voxel_size, vps, block_index, tsdf_distance, tsdf_weight, min_weight=1.0):
    """VoxgraphSubmap::findIsosurfaceVertices -> (xyz[n,3], distance[n], weight[n])."""
    L = lib()
    L.orc_isosurface_points.argtypes = [C.c_float, C.c_int, C.c_int, c_i32p, c_f32p, c_f32p,
                                        C.c_float, c_f32p, c_f32p, c_f32p]
    L.orc_isosurface_points.restype = C.c_int64
    bi = np.ascontiguousarray(block_index, np.int32).reshape(-1, 3)
    td, tw = _f32(tsdf_distance), _f32(tsdf_weight)
    args = (float(np.float32(voxel_size)), vps, bi.shape[0], _p(bi, c_i32p), _p(td, c_f32p),
            _p(tw, c_f32p), float(min_weight))
    n = L.orc_isosurface_points(*args, None, None, None)
    if n < 0:
        raise MemoryError("orc_isosurface_points")
    xyz, d, w = np.zeros((n, 3), np.float32), np.zeros(n, np.float32), np.zeros(n, np.float32)
    L.orc_isosurface_points(*args, _p(xyz, c_f32p), _p(d, c_f32p), _p(w, c_f32p))
    return xyz, d, w


def tsdf_replay_check(config, voxel_size, vps, T_G_C, points_C, colors, freespace_points, offsets, sets_pre, sets_post,
                      layer_pre, layer_post, trace):
    """oracle/tsdf_replay.h: is `trace` (one racing scan's event log) a legal interleaving of the sequential integrator's
    steps?  offsets = (start, observed) offsets of the scan's values; sets_* = (start set, observed set); layer_* = the
    four arrays of a layer download.  Returns the report as a dict (errors == 0: legal)."""
    u64p = C.POINTER(C.c_uint64)
    T = _f32(T_G_C)
    pts = _f32(points_C).reshape(-1, 3)
    col = None if colors is None else np.ascontiguousarray(colors, np.uint8).reshape(-1, 4)
    keep = []

    def layer(arrs):
        bi, d, w, c = arrs
        bi = np.ascontiguousarray(bi, np.int32).reshape(-1, 3)
        d, w = _f32(d), _f32(w)
        c = np.ascontiguousarray(c, np.uint8)
        keep.extend((bi, d, w, c))
        return ReplayLayer(len(bi), _p(bi, c_i32p), _p(d, c_f32p), _p(w, c_f32p), _p(c, c_u8p))

    pre, post = layer(layer_pre), layer(layer_post)
    sets = [np.ascontiguousarray(x, np.uint64) for x in (*sets_pre, *sets_post)]
    tr = np.ascontiguousarray(trace, np.uint64)
    rep = ReplayReport()
    _tsdf_lib().orc_tsdf_replay_check(C.byref(config), float(np.float32(voxel_size)), int(vps), _p(T, c_f32p), _p(pts, c_f32p),
                                      _p(col, c_u8p), pts.shape[0], int(freespace_points), int(offsets[0]), int(offsets[1]),
                                      _p(sets[0], u64p), _p(sets[2], u64p), _p(sets[1], u64p), _p(sets[3], u64p),
                                      C.byref(pre), C.byref(post), _p(tr, u64p), len(tr), C.byref(rep))
    return rep.as_dict()
