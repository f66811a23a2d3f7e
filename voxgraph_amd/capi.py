"""ctypes view of libvoxgraph_amd.so (include/voxgraph_amd.h).

This module only declares the C ABI and wraps handles; all arithmetic happens in
the HIP library.  There is no fallback: if the shared object is missing the
import fails, and without a gfx950 device Context() raises.
"""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
# VGX_LIB: an alternative build of the same library (A/B scripts under profiles/ only)
LIB_PATH = os.environ.get("VGX_LIB") or os.path.join(_HERE, "lib", "libvoxgraph_amd.so")
# test / benchmark tooling (include/voxgraph_amd_bench.h), built next to it from csrc/bench/
BENCH_LIB_PATH = os.path.join(os.path.dirname(LIB_PATH), os.path.basename(LIB_PATH).replace("libvoxgraph_amd", "libvoxgraph_amd_bench", 1))

OK = 0
EVALUATE_FALSE = 1
ERR_INVALID, ERR_HIP, ERR_NOMEM, ERR_UNSUPPORTED, ERR_NO_DEVICE = -1, -2, -3, -4, -5

BRICKS_APRON, BRICKS_QUAD = 0, 1
SAMPLING_BRICKS_SAME, SAMPLING_BRICKS_QUAD = 0, 1
POINTS_ISOSURFACE = 0
POINTS_VOXELS = 1
POINTS_KEEP_ORDER = 0
POINTS_SORT_MORTON = 1

NORMAL_SIZE = 45

vp = C.c_void_p
f32p = C.POINTER(C.c_float)
f64p = C.POINTER(C.c_double)
i32p = C.POINTER(C.c_int32)
i64p = C.POINTER(C.c_int64)
u8p = C.POINTER(C.c_uint8)


class RegConfig(C.Structure):
    """vgx_reg_config == RegistrationCostFunction::Config (registration_cost_function.h:17-41)."""
    _fields_ = [("registration_point_type", C.c_int32), ("sampling_ratio", C.c_float),
                ("no_correspondence_cost", C.c_double), ("use_esdf_distance", C.c_int32),
                ("sampler_seed", C.c_uint32)]


class EsdfConfig(C.Structure):
    """vgx_esdf_config == voxblox::EsdfIntegrator::Config (the fields used in batch mode)."""
    _fields_ = [("max_distance_m", C.c_float), ("min_distance_m", C.c_float),
                ("default_distance_m", C.c_float), ("min_diff_m", C.c_float),
                ("min_weight", C.c_float), ("num_buckets", C.c_int32)]


class MapFileSubmapInfo(C.Structure):
    _fields_ = [("id", C.c_int64), ("T_M_S", C.c_double * 7), ("voxel_size", C.c_double),
                ("voxels_per_side", C.c_int32), ("n_tsdf_blocks", C.c_int32),
                ("n_esdf_blocks", C.c_int32), ("layer_is_esdf", C.c_int32)]


class MapFileSubmapData(C.Structure):
    _fields_ = [("id", C.c_int64), ("T_M_S", C.c_double * 7), ("n_blocks", C.c_int32),
                ("block_index", C.POINTER(C.c_int32)), ("tsdf_distance", C.POINTER(C.c_float)),
                ("tsdf_weight", C.POINTER(C.c_float)), ("tsdf_rgba", C.POINTER(C.c_uint8)),
                ("esdf_distance", C.POINTER(C.c_float)), ("esdf_observed", C.POINTER(C.c_uint8))]


class TsdfConfig(C.Structure):
    """vgx_tsdf_config == voxblox::TsdfIntegratorBase::Config (the fields that matter on a GPU)."""
    _fields_ = [("default_truncation_distance", C.c_float), ("max_weight", C.c_float),
                ("voxel_carving_enabled", C.c_int32), ("min_ray_length_m", C.c_float),
                ("max_ray_length_m", C.c_float), ("use_const_weight", C.c_int32),
                ("allow_clear", C.c_int32), ("use_weight_dropoff", C.c_int32),
                ("use_sparsity_compensation_factor", C.c_int32),
                ("sparsity_compensation_factor", C.c_float),
                ("start_voxel_subsampling_factor", C.c_float),
                ("max_consecutive_ray_collisions", C.c_int32),
                ("clear_checks_every_n_frames", C.c_int32), ("enable_anti_grazing", C.c_int32),
                ("deterministic", C.c_int32), ("integration_order", C.c_int32)]


TSDF_ORDER_MIXED, TSDF_ORDER_SORTED = 0, 1

# every symbol include/voxgraph_amd.h declares: name -> (restype, argtypes)
SIGNATURES = {
    "vgx_ctx_create": (C.c_int, [C.c_int, C.POINTER(vp)]),
    "vgx_ctx_destroy": (C.c_int, [vp]),
    "vgx_last_error": (C.c_char_p, [vp]),
    "vgx_ctx_set_stream": (C.c_int, [vp, vp]),
    "vgx_ctx_get_stream": (vp, [vp]),
    "vgx_ctx_set_tsdf_stream": (C.c_int, [vp, vp]),
    "vgx_ctx_get_tsdf_stream": (vp, [vp]),
    "vgx_ctx_synchronize": (C.c_int, [vp]),
    "vgx_ctx_synchronize_tsdf": (C.c_int, [vp]),
    "vgx_ctx_tsdf_wait_for_stream": (C.c_int, [vp, vp]),
    "vgx_ctx_stream_priorities": (C.c_int, [vp]),
    "vgx_ctx_set_brick_layout": (C.c_int, [vp, C.c_int32]),
    "vgx_ctx_set_sampling_bricks": (C.c_int, [vp, C.c_int32]),
    "vgx_ctx_timer_start": (C.c_int, [vp]),
    "vgx_ctx_timer_stop": (C.c_int, [vp, f32p]),
    "vgx_submap_create": (C.c_int, [vp, C.c_int32, C.c_float, C.c_int32, C.c_int32, i32p, f32p,
                                    f32p, f32p, u8p, C.POINTER(vp)]),
    "vgx_submap_destroy": (C.c_int, [vp]),
    "vgx_submap_id": (C.c_int32, [vp]),
    "vgx_submap_num_blocks": (C.c_int32, [vp]),
    "vgx_submap_set_points": (C.c_int, [vp, C.c_int32, C.c_int64, f32p, f32p, f32p, C.c_uint32]),
    "vgx_submap_extract_voxel_points": (C.c_int, [vp, C.c_double, C.c_double, C.c_int32, i64p]),
    "vgx_submap_extract_isosurface_points": (C.c_int, [vp, C.c_double, i64p]),
    "vgx_submap_num_points": (C.c_int64, [vp, C.c_int32]),
    "vgx_submap_point_order": (C.c_int, [vp, C.c_int32, i64p]),
    "vgx_submap_download_points": (C.c_int, [vp, C.c_int32, f32p, f32p, f32p]),
    "vgx_esdf_config_default": (None, [C.POINTER(EsdfConfig)]),
    "vgx_submap_generate_esdf": (C.c_int, [vp, C.POINTER(EsdfConfig), i32p]),
    "vgx_submap_from_tsdf_layer": (C.c_int, [vp, vp, C.c_int32, C.POINTER(vp)]),
    "vgx_submap_release_raw_layers": (C.c_int, [vp]),
    "vgx_submap_download_layers": (C.c_int, [vp, f32p, f32p, f32p, u8p]),
    "vgx_submap_block_index": (C.c_int, [vp, i32p]),
    "vgx_reg_config_default": (None, [C.POINTER(RegConfig)]),
    "vgx_reg_create": (C.c_int, [vp, vp, vp, C.POINTER(RegConfig), C.POINTER(vp)]),
    "vgx_reg_destroy": (C.c_int, [vp]),
    "vgx_reg_num_residuals": (C.c_int64, [vp]),
    "vgx_reg_evaluate": (C.c_int, [vp, f64p, f64p, f64p, f64p, f64p]),
    "vgx_reg_evaluate_device_f32": (C.c_int, [vp, f64p, f64p, vp, vp, vp]),
    "vgx_reg_batch_create": (C.c_int, [vp, C.c_int32, C.POINTER(vp), i32p, i32p, C.c_int32,
                                       C.POINTER(vp)]),
    "vgx_reg_batch_destroy": (C.c_int, [vp]),
    "vgx_reg_batch_num_residuals": (C.c_int64, [vp]),
    "vgx_reg_batch_row_offsets": (C.c_int, [vp, i64p]),
    "vgx_reg_batch_evaluate_points": (C.c_int, [vp, f64p, C.c_int32, vp, vp, vp, i32p]),
    "vgx_reg_batch_evaluate_points_f64": (C.c_int, [vp, f64p, C.c_int32, vp, vp, vp, i32p]),
    "vgx_reg_batch_evaluate_rows_f64": (C.c_int, [vp, f64p, C.c_int32, C.c_int32, C.c_int32, i32p]),
    "vgx_reg_batch_fetch_rows_f64": (C.c_int, [vp, C.c_int32, f64p, f64p, f64p]),
    "vgx_reg_batch_evaluate_cost": (C.c_int, [vp, f64p, C.c_int32, vp, f64p, i32p]),
    "vgx_reg_batch_blocked_layout": (C.c_int, [vp, C.POINTER(C.c_int64), i32p, C.POINTER(C.c_int64)]),
    "vgx_reg_batch_evaluate_points_blocked": (C.c_int, [vp, f64p, C.c_int32, vp, i32p]),
    "vgx_reg_batch_choose_outputs": (C.c_int, [vp, f64p, C.c_int32, C.c_int32, C.POINTER(vp), C.POINTER(vp), C.POINTER(vp),
                                               C.c_int32, i32p, C.POINTER(C.c_float), C.POINTER(C.c_float)]),
    "vgx_reg_batch_alloc_outputs": (C.c_int, [vp, f64p, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.POINTER(vp), C.POINTER(vp),
                                              C.POINTER(vp), C.POINTER(C.c_float)]),
    "vgx_reg_batch_free_outputs": (C.c_int, [vp, vp, vp, vp]),
    "vgx_reg_batch_evaluate_normal": (C.c_int, [vp, f64p, C.c_int32, vp, f64p, i32p]),
    "vgx_reg_batch_count_live": (C.c_int, [vp, f64p, C.c_int32, i64p, i64p]),
    "vgx_reg_batch_launch_order": (C.c_int, [vp, C.c_int32, i32p]),
    "vgx_reg_batch_count_live_each": (C.c_int, [vp, f64p, C.c_int32, i64p]),
    "vgx_reg_batch_assemble": (C.c_int, [vp, vp, C.c_int32, vp, C.c_int32]),
    "vgx_reg_fused_size": (C.c_int64, [C.c_int32, C.c_int32]),
    "vgx_reg_batch_scatter_normal": (C.c_int, [vp, vp, vp, C.c_int32]),
    "vgx_reg_batch_brick_layout": (C.c_int32, [vp]),
    "vgx_reg_assembler_create": (C.c_int, [vp, C.c_int32, i32p, C.POINTER(vp)]),
    "vgx_reg_assembler_assemble": (C.c_int, [vp, vp, C.c_int32, vp]),
    "vgx_reg_assembler_destroy": (C.c_int, [vp]),
    "vgx_lpt_shards": (C.c_int, [C.c_int32, i64p, C.c_int32, i32p]),
    "vgx_contiguous_shards": (C.c_int, [C.c_int32, i64p, C.c_int32, i32p]),
    "vgx_reg_multi_create": (C.c_int, [C.c_int32, C.POINTER(vp), C.c_int32, C.POINTER(vp), i32p, C.POINTER(vp)]),
    "vgx_reg_multi_destroy": (C.c_int, [vp]),
    "vgx_reg_multi_num_shards": (C.c_int32, [vp]),
    "vgx_reg_multi_set_reduction": (C.c_int, [vp, C.c_int32]),
    "vgx_reg_multi_shard_of": (C.c_int, [vp, i32p]),
    "vgx_reg_multi_evaluate_fused": (C.c_int, [vp, f64p, C.c_int32, f64p, i32p]),
    "vgx_reg_multi_evaluate_normal": (C.c_int, [vp, f64p, C.c_int32, f64p, i32p]),
    "vgx_reg_multi_evaluate_cost": (C.c_int, [vp, f64p, C.c_int32, f64p, i32p]),
    "vgx_reg_compress_normal": (C.c_int, [f64p, f64p, f64p]),
    "vgx_submap_surface_obb": (C.c_int, [vp, f32p, f32p]),
    "vgx_submap_mission_surface_aabb": (C.c_int, [vp, f64p, f32p, f32p]),
    "vgx_find_overlapping_pairs": (C.c_int, [vp, C.c_int32, C.POINTER(vp), f64p, i32p, C.c_int32, i32p]),
    "vgx_tsdf_config_default": (None, [C.POINTER(TsdfConfig)]),
    "vgx_tsdf_layer_create": (C.c_int, [vp, C.c_float, C.c_int32, i32p, i32p, C.c_int32, C.POINTER(vp)]),
    "vgx_tsdf_layer_destroy": (C.c_int, [vp]),
    "vgx_tsdf_layer_stats": (C.c_int, [vp, i32p, i64p]),
    "vgx_tsdf_layer_reserve": (C.c_int, [vp, f32p, C.c_float]),
    "vgx_tsdf_layer_clear_dropped": (C.c_int, [vp]),
    "vgx_tsdf_layer_growths": (C.c_int64, [vp]),
    "vgx_tsdf_layer_download": (C.c_int, [vp, i32p, f32p, f32p, u8p]),
    "vgx_tsdf_layer_upload": (C.c_int, [vp, C.c_int32, i32p, f32p, f32p, u8p]),
    "vgx_tsdf_integrator_create": (C.c_int, [vp, C.POINTER(TsdfConfig), vp, C.POINTER(vp)]),
    "vgx_tsdf_integrator_destroy": (C.c_int, [vp]),
    "vgx_tsdf_integrator_set_layer": (C.c_int, [vp, vp]),
    "vgx_tsdf_integrator_set_cloud_width": (C.c_int, [vp, C.c_int32]),
    "vgx_tsdf_integrate": (C.c_int, [vp, f32p, f32p, u8p, C.c_int64, C.c_int32, i64p]),
    "vgx_tsdf_integrate_device": (C.c_int, [vp, f32p, vp, vp, C.c_int64, C.c_int32, i64p]),
    "vgx_tsdf_integrate_merged": (C.c_int, [vp, f32p, f32p, u8p, C.c_int64, C.c_int32, i64p]),
    "vgx_tsdf_integrate_merged_device": (C.c_int, [vp, f32p, vp, vp, C.c_int64, C.c_int32, i64p]),
    "vgx_map_file_open": (C.c_int, [C.c_char_p, C.c_int32, C.POINTER(vp)]),
    "vgx_map_file_close": (C.c_int, [vp]),
    "vgx_map_file_last_error": (C.c_char_p, [vp]),
    "vgx_map_file_num_submaps": (C.c_int32, [vp]),
    "vgx_map_file_get_submap_info": (C.c_int, [vp, C.c_int32, C.POINTER(MapFileSubmapInfo)]),
    "vgx_map_file_read_submap": (C.c_int, [vp, C.c_int32, i32p, f32p, f32p, u8p, f32p, u8p]),
    "vgx_map_file_load_submap": (C.c_int, [vp, vp, C.c_int32, C.POINTER(vp)]),
    "vgx_map_file_write": (C.c_int, [C.c_char_p, C.c_int32, C.c_double, C.c_int32, C.c_int32,
                                     C.POINTER(MapFileSubmapData)]),
}

# every symbol include/voxgraph_amd_bench.h declares (libvoxgraph_amd_bench.so: test and benchmark tooling)
BENCH_SIGNATURES = {
    "vgx_synth_city_submap": (C.c_int, [vp, C.c_int32, C.c_float, C.c_int32, i32p, i32p, C.c_float,
                                        C.c_float, C.c_float, f64p, C.c_uint32, C.c_int32,
                                        C.POINTER(vp)]),
    "vgx_synth_city_scan": (C.c_int, [vp, f64p, C.c_int32, C.c_int32, C.c_float, C.c_float,
                                      C.c_uint32, vp]),
    "vgx_bench_atomic_roundtrip": (C.c_int, [vp, C.c_int64, C.c_int32, C.c_int32, f32p]),
    "vgx_bench_stream_ceiling": (C.c_int, [vp, vp, C.c_int64, vp, C.c_int64, C.c_int32, f32p]),
    "vgx_bench_alloc_scattered": (C.c_int, [vp, C.c_int64, C.c_int64, C.c_uint32, C.POINTER(vp)]),
    "vgx_bench_free_scattered": (C.c_int, [vp, vp]),
    "vgx_tsdf_integrator_walk_stats": (C.c_int, [vp, i64p]),
    "vgx_tsdf_integrator_read_trace": (C.c_int, [vp, i64p, C.c_int64, i64p, i64p]),
    "vgx_tsdf_integrator_set_speculation": (C.c_int, [vp, C.c_int32, C.c_int64]),
    "vgx_tsdf_integrator_set_event_trace": (C.c_int, [vp, C.c_int64]),
    "vgx_tsdf_integrator_read_event_trace": (C.c_int, [vp, C.POINTER(C.c_uint64), C.c_int64, i64p, i64p]),
    "vgx_tsdf_integrator_download_sets": (C.c_int, [vp, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64), i64p]),
}

_lib = None


def _share_hip_runtime_with_torch():
    """PyTorch wheels bundle their own libamdhip64.so.7 / libhsa-runtime64; two HIP
    runtimes in one process cannot both open the GPU.  If torch is installed, load
    ITS runtime first (RTLD_GLOBAL): libvoxgraph_amd.so's DT_NEEDED libamdhip64.so.7
    then resolves to the same copy by SONAME, and a later `import torch` reuses it."""
    import importlib.util
    import sys
    if "torch" in sys.modules:
        return
    spec = importlib.util.find_spec("torch")
    if spec is None or not spec.submodule_search_locations:
        return
    cand = os.path.join(list(spec.submodule_search_locations)[0], "lib", "libamdhip64.so")
    if os.path.exists(cand):
        C.CDLL(cand, mode=C.RTLD_GLOBAL)


def load():
    """dlopen libvoxgraph_amd.so and bind every declared symbol (loud on failure)."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise ImportError(
                f"{LIB_PATH} is missing: build it with `python -c 'import __graft_entry__ as g; "
                "g.build()'` (hipcc --offload-arch=gfx950). There is no CPU fallback.")
        _share_hip_runtime_with_torch()
        lib = C.CDLL(LIB_PATH, mode=C.RTLD_GLOBAL)   # (the tooling library resolves its undefined symbols in it)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(lib, name)     # AttributeError if the symbol is not exported
            fn.restype = res
            fn.argtypes = args
        _lib = _Libraries(lib)
    return _lib


class _Libraries:
    """The product library, plus -- bound at the first use of one of its symbols -- the tooling library: tests and bench.py
    call both through `ctx.lib`; an integration loads libvoxgraph_amd.so alone."""

    def __init__(self, product):
        self.product = product
        self.tooling = None

    def __getattr__(self, name):
        if name in SIGNATURES:
            return getattr(self.product, name)
        if name in BENCH_SIGNATURES:
            if self.tooling is None:
                if not os.path.exists(BENCH_LIB_PATH):
                    raise ImportError(f"{BENCH_LIB_PATH} is missing: build it with __graft_entry__.build()")
                tooling = C.CDLL(BENCH_LIB_PATH)
                for sym, (res, args) in BENCH_SIGNATURES.items():
                    fn = getattr(tooling, sym)
                    fn.restype = res
                    fn.argtypes = args
                self.tooling = tooling
            return getattr(self.tooling, name)
        raise AttributeError(name)


class VgxError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__(f"vgx error {code}: {msg}")
        self.code = code


def _ptr(a, t):
    return None if a is None else a.ctypes.data_as(t)


def _f32(a):
    return None if a is None else np.ascontiguousarray(a, dtype=np.float32)


def _f64(a):
    return np.ascontiguousarray(a, dtype=np.float64)


class Context:
    def __init__(self, device=0):
        self.lib = load()
        h = vp()
        rc = self.lib.vgx_ctx_create(device, C.byref(h))
        if rc != OK:
            raise VgxError(rc, self.lib.vgx_last_error(None).decode())
        self.h = h

    def check(self, rc):
        if rc < 0:
            raise VgxError(rc, self.lib.vgx_last_error(self.h).decode())
        return rc

    def set_stream(self, stream_ptr):
        self.check(self.lib.vgx_ctx_set_stream(self.h, vp(stream_ptr)))

    def get_stream(self):
        """the hipStream_t (as an int) the library launches on"""
        return int(self.lib.vgx_ctx_get_stream(self.h) or 0)

    def set_tsdf_stream(self, stream_ptr):
        """the TSDF side's stream (layers, integrators, scans); None restores the context's own"""
        self.check(self.lib.vgx_ctx_set_tsdf_stream(self.h, vp(stream_ptr) if stream_ptr else None))

    def get_tsdf_stream(self):
        return int(self.lib.vgx_ctx_get_tsdf_stream(self.h) or 0)

    def synchronize(self):
        self.check(self.lib.vgx_ctx_synchronize(self.h))

    def synchronize_tsdf(self):
        """waits for the TSDF side alone (a scan is in), whatever the registration side has queued"""
        self.check(self.lib.vgx_ctx_synchronize_tsdf(self.h))

    def tsdf_wait_for_stream(self, producer_stream=None):
        """the TSDF stream waits on the device for what producer_stream (None: the registration stream) holds now"""
        self.check(self.lib.vgx_ctx_tsdf_wait_for_stream(self.h, vp(int(producer_stream)) if producer_stream else None))

    def stream_priorities(self):
        return bool(self.lib.vgx_ctx_stream_priorities(self.h))

    def set_brick_layout(self, layout):
        """BRICKS_APRON (default) or BRICKS_QUAD, for the submaps created from now on"""
        self.check(self.lib.vgx_ctx_set_brick_layout(self.h, int(layout)))

    def set_sampling_bricks(self, mode):
        """SAMPLING_BRICKS_QUAD (default: all-sampling batches read quad bricks made on demand) or _SAME"""
        self.check(self.lib.vgx_ctx_set_sampling_bricks(self.h, int(mode)))

    def timer_start(self):
        self.check(self.lib.vgx_ctx_timer_start(self.h))

    def timer_stop(self):
        ms = C.c_float()
        self.check(self.lib.vgx_ctx_timer_stop(self.h, C.byref(ms)))
        return ms.value

    def close(self):
        if self.h:
            self.lib.vgx_ctx_destroy(self.h)
            self.h = None


class Submap:
    """A finished VoxgraphSubmap resident on the GPU."""

    def __init__(self, ctx, submap_id, voxel_size, vps, block_index, tsdf_distance=None,
                 tsdf_weight=None, esdf_distance=None, esdf_observed=None):
        self.ctx = ctx
        bi = np.ascontiguousarray(block_index, dtype=np.int32).reshape(-1, 3)
        td, tw, ed = _f32(tsdf_distance), _f32(tsdf_weight), _f32(esdf_distance)
        eo = None if esdf_observed is None else np.ascontiguousarray(esdf_observed, np.uint8)
        h = vp()
        ctx.check(ctx.lib.vgx_submap_create(ctx.h, submap_id, float(voxel_size), vps, bi.shape[0],
                                            _ptr(bi, i32p), _ptr(td, f32p), _ptr(tw, f32p),
                                            _ptr(ed, f32p), _ptr(eo, u8p), C.byref(h)))
        self.h = h

    @classmethod
    def synth_city(cls, ctx, submap_id, voxel_size, vps, block_min, block_dims, truncation,
                   esdf_max, tsdf_weight, true_pose, seed, build_tsdf_grid=False):
        """Benchmark tooling: analytic city scene generated on the device."""
        self = cls.__new__(cls)
        self.ctx = ctx
        bmin = np.ascontiguousarray(block_min, np.int32)
        bdim = np.ascontiguousarray(block_dims, np.int32)
        pose = _f64(true_pose)
        h = vp()
        ctx.check(ctx.lib.vgx_synth_city_submap(
            ctx.h, submap_id, float(voxel_size), vps, _ptr(bmin, i32p), _ptr(bdim, i32p),
            float(truncation), float(esdf_max), float(tsdf_weight), _ptr(pose, f64p), seed,
            int(build_tsdf_grid), C.byref(h)))
        self.h = h
        return self

    @classmethod
    def from_tsdf_layer(cls, ctx, layer, submap_id):
        """finishSubmap() hand-off on the device (no host round trip)."""
        self = cls.__new__(cls)
        self.ctx = ctx
        h = vp()
        ctx.check(ctx.lib.vgx_submap_from_tsdf_layer(ctx.h, layer.h, submap_id, C.byref(h)))
        self.h = h
        return self

    def generate_esdf(self, config=None):
        """cblox::TsdfEsdfSubmap::generateEsdf() on the device; returns global passes used."""
        n = C.c_int32()
        self.ctx.check(self.ctx.lib.vgx_submap_generate_esdf(
            self.h, C.byref(config) if config is not None else None, C.byref(n)))
        return n.value

    def num_blocks(self):
        return self.ctx.lib.vgx_submap_num_blocks(self.h)

    def block_index(self):
        bi = np.zeros((self.num_blocks(), 3), np.int32)
        self.ctx.check(self.ctx.lib.vgx_submap_block_index(self.h, _ptr(bi, i32p)))
        return bi

    def download_layers(self, vps):
        n = self.num_blocks()
        td = np.zeros((n, vps ** 3), np.float32)
        tw = np.zeros((n, vps ** 3), np.float32)
        ed = np.zeros((n, vps ** 3), np.float32)
        eo = np.zeros((n, vps ** 3), np.uint8)
        self.ctx.check(self.ctx.lib.vgx_submap_download_layers(
            self.h, _ptr(td, f32p), _ptr(tw, f32p), _ptr(ed, f32p), _ptr(eo, u8p)))
        return td, tw, ed, eo

    def set_points(self, point_type, xyz, distance, weight, flags=POINTS_KEEP_ORDER):
        xyz = _f32(xyz).reshape(-1, 3)
        d, w = _f32(distance), _f32(weight)
        self.ctx.check(self.ctx.lib.vgx_submap_set_points(
            self.h, point_type, xyz.shape[0], _ptr(xyz, f32p), _ptr(d, f32p), _ptr(w, f32p), flags))

    def extract_voxel_points(self, min_voxel_weight=1.0, max_voxel_distance=0.3,
                             use_esdf_distance=True):
        n = C.c_int64()
        self.ctx.check(self.ctx.lib.vgx_submap_extract_voxel_points(
            self.h, min_voxel_weight, max_voxel_distance, int(use_esdf_distance), C.byref(n)))
        return n.value

    def extract_isosurface_points(self, min_voxel_weight=1.0):
        n = C.c_int64()
        self.ctx.check(self.ctx.lib.vgx_submap_extract_isosurface_points(
            self.h, min_voxel_weight, C.byref(n)))
        return n.value

    def surface_obb(self):
        mn, mx = np.zeros(3, np.float32), np.zeros(3, np.float32)
        self.ctx.check(self.ctx.lib.vgx_submap_surface_obb(self.h, _ptr(mn, f32p), _ptr(mx, f32p)))
        return mn, mx

    def mission_surface_aabb(self, pose):
        mn, mx = np.zeros(3, np.float32), np.zeros(3, np.float32)
        self.ctx.check(self.ctx.lib.vgx_submap_mission_surface_aabb(
            self.h, _ptr(_f64(pose), f64p), _ptr(mn, f32p), _ptr(mx, f32p)))
        return mn, mx

    def num_points(self, point_type):
        return self.ctx.lib.vgx_submap_num_points(self.h, point_type)

    def point_order(self, point_type):
        n = self.num_points(point_type)
        order = np.zeros(max(n, 0), np.int64)
        self.ctx.check(self.ctx.lib.vgx_submap_point_order(self.h, point_type, _ptr(order, i64p)))
        return order

    def download_points(self, point_type):
        n = self.num_points(point_type)
        xyz = np.zeros((n, 3), np.float32)
        d = np.zeros(n, np.float32)
        w = np.zeros(n, np.float32)
        self.ctx.check(self.ctx.lib.vgx_submap_download_points(
            self.h, point_type, _ptr(xyz, f32p), _ptr(d, f32p), _ptr(w, f32p)))
        return xyz, d, w

    def release_raw_layers(self):
        self.ctx.check(self.ctx.lib.vgx_submap_release_raw_layers(self.h))

    def destroy(self):
        if self.h:
            self.ctx.lib.vgx_submap_destroy(self.h)
            self.h = None


def default_config(**kw):
    cfg = RegConfig()
    load().vgx_reg_config_default(C.byref(cfg))
    for k, v in kw.items():
        if not hasattr(cfg, k):
            raise AttributeError(k)
        setattr(cfg, k, v)
    return cfg


class RegistrationCostFunction:
    """Mirror of voxgraph::RegistrationCostFunction over the C ABI.

    Evaluate(parameters, residuals, jacobians) follows ceres::CostFunction::Evaluate
    (registration_cost_function.h:47-48): parameters = [ref_pose[4], read_pose[4]],
    residuals = f64[N], jacobians = None or [f64[N,4] or None, f64[N,4] or None].
    Returns True/False like the reference.
    """

    def __init__(self, ctx, reference_submap, reading_submap, config=None):
        self.ctx = ctx
        self.config = config if config is not None else default_config()
        h = vp()
        ctx.check(ctx.lib.vgx_reg_create(ctx.h, reference_submap.h, reading_submap.h,
                                         C.byref(self.config), C.byref(h)))
        self.h = h
        self._keep = (reference_submap, reading_submap)

    def num_residuals(self):
        return self.ctx.lib.vgx_reg_num_residuals(self.h)

    def Evaluate(self, parameters, residuals, jacobians):
        ref = _f64(parameters[0])
        read = _f64(parameters[1])
        jr = je = None
        if jacobians is not None:
            jr, je = jacobians[0], jacobians[1]
        for a in (residuals, jr, je):
            assert a is None or (a.dtype == np.float64 and a.flags.c_contiguous)
        rc = self.ctx.check(self.ctx.lib.vgx_reg_evaluate(
            self.h, _ptr(ref, f64p), _ptr(read, f64p), _ptr(residuals, f64p), _ptr(jr, f64p),
            _ptr(je, f64p)))
        return rc == OK

    def evaluate_device_f32(self, ref_pose, read_pose, d_residuals, d_jac_ref, d_jac_read):
        """Device pointers (ints); asynchronous on the context's stream."""
        rc = self.ctx.check(self.ctx.lib.vgx_reg_evaluate_device_f32(
            self.h, _ptr(_f64(ref_pose), f64p), _ptr(_f64(read_pose), f64p), vp(d_residuals),
            vp(d_jac_ref) if d_jac_ref else None, vp(d_jac_read) if d_jac_read else None))
        return rc == OK

    def destroy(self):
        if self.h:
            self.ctx.lib.vgx_reg_destroy(self.h)
            self.h = None


class RegistrationBatch:
    """All registration constraints of a pose graph, evaluated in one launch."""

    def __init__(self, ctx, cost_functions, node_pair, global_index=None, n_global=None):
        self.ctx = ctx
        self.n = len(cost_functions)
        arr = (vp * max(self.n, 1))(*[cf.h for cf in cost_functions])
        np_pair = np.ascontiguousarray(node_pair, dtype=np.int32).reshape(-1, 2)
        gi = None if global_index is None else np.ascontiguousarray(global_index, np.int32)
        h = vp()
        ctx.check(ctx.lib.vgx_reg_batch_create(
            ctx.h, self.n, arr, _ptr(np_pair, i32p), _ptr(gi, i32p),
            self.n if n_global is None else n_global, C.byref(h)))
        self.h = h
        self.n_global = self.n if n_global is None else n_global
        self._keep = list(cost_functions)

    def num_residuals(self):
        return self.ctx.lib.vgx_reg_batch_num_residuals(self.h)

    def row_offsets(self):
        ro = np.zeros(self.n + 1, np.int64)
        self.ctx.check(self.ctx.lib.vgx_reg_batch_row_offsets(self.h, _ptr(ro, i64p)))
        return ro

    def evaluate_points(self, poses, d_residuals, d_jac_ref, d_jac_read):
        poses = _f64(poses).reshape(-1, 4)
        status = np.zeros(max(self.n, 1), np.int32)
        self.ctx.check(self.ctx.lib.vgx_reg_batch_evaluate_points(
            self.h, _ptr(poses, f64p), poses.shape[0], vp(d_residuals),
            vp(d_jac_ref) if d_jac_ref else None, vp(d_jac_read) if d_jac_read else None,
            _ptr(status, i32p)))
        return status[:self.n]

    def evaluate_points_f64(self, poses, d_residuals, d_jac_ref, d_jac_read):
        """vgx_reg_batch_evaluate_points_f64: the same rows as f64 (Ceres' own types) into device arrays"""
        poses = _f64(poses).reshape(-1, 4)
        status = np.zeros(max(self.n, 1), np.int32)
        self.ctx.check(self.ctx.lib.vgx_reg_batch_evaluate_points_f64(
            self.h, _ptr(poses, f64p), poses.shape[0], vp(d_residuals),
            vp(d_jac_ref) if d_jac_ref else None, vp(d_jac_read) if d_jac_read else None,
            _ptr(status, i32p)))
        return status[:self.n]

    def evaluate_rows_f64(self, poses, want_jac_ref=True, want_jac_read=True):
        """vgx_reg_batch_evaluate_rows_f64: f64 rows kept by the batch -> status per constraint"""
        poses = _f64(poses).reshape(-1, 4)
        status = np.zeros(max(self.n, 1), np.int32)
        self.ctx.check(self.ctx.lib.vgx_reg_batch_evaluate_rows_f64(self.h, _ptr(poses, f64p), poses.shape[0], int(want_jac_ref),
                                                                    int(want_jac_read), _ptr(status, i32p)))
        return status[:self.n]

    def fetch_rows_f64(self, c, n, want_jac_ref=True, want_jac_read=True):
        """vgx_reg_batch_fetch_rows_f64: constraint c's slice (n = its residuals) -> (residuals, jac_ref or None, jac_read or None)"""
        r = np.full(n, np.nan)
        jo = np.full((n, 4), np.nan) if want_jac_ref else None
        je = np.full((n, 4), np.nan) if want_jac_read else None
        self.ctx.check(self.ctx.lib.vgx_reg_batch_fetch_rows_f64(self.h, int(c), _ptr(r, f64p), _ptr(jo, f64p), _ptr(je, f64p)))
        return r, jo, je

    def blocked_layout(self):
        """vgx_reg_batch_blocked_layout -> (bytes, rows per block, first_block [n + 1])"""
        nbytes, rows = C.c_int64(), C.c_int32()
        first = np.zeros(self.n + 1, np.int64)
        self.ctx.check(self.ctx.lib.vgx_reg_batch_blocked_layout(self.h, C.byref(nbytes), C.byref(rows),
                                                                 first.ctypes.data_as(C.POINTER(C.c_int64))))
        return int(nbytes.value), int(rows.value), first

    def evaluate_points_blocked(self, poses, d_blocks):
        poses = _f64(poses).reshape(-1, 4)
        status = np.zeros(max(self.n, 1), np.int32)
        self.ctx.check(self.ctx.lib.vgx_reg_batch_evaluate_points_blocked(self.h, _ptr(poses, f64p), poses.shape[0], vp(d_blocks),
                                                                          _ptr(status, i32p)))
        return status[:self.n]

    def choose_outputs(self, poses, d_residuals, d_jac_ref, d_jac_read, launches=3):
        """vgx_reg_batch_choose_outputs: lists of candidate device pointers (d_jac_ref / d_jac_read may be None) ->
        (chosen [3], ms per launch of the chosen combination, ms of every trial)"""
        poses = _f64(poses).reshape(-1, 4)
        n = len(d_residuals)
        arr = lambda ps: (vp * n)(*[vp(int(x)) for x in ps]) if ps is not None else None
        a_r, a_jr, a_je = arr(d_residuals), arr(d_jac_ref), arr(d_jac_read)
        chosen = np.zeros(3, np.int32)
        ms = C.c_float()
        trials = (C.c_float * (4 * n))()
        self.ctx.check(self.ctx.lib.vgx_reg_batch_choose_outputs(
            self.h, _ptr(poses, f64p), poses.shape[0], n, a_r, a_jr, a_je, int(launches), _ptr(chosen, i32p),
            C.byref(ms), trials))
        return [int(x) for x in chosen], float(ms.value), [float(x) for x in trials]

    def alloc_outputs(self, poses, n_candidates=4, want_jac_ref=True, want_jac_read=True):
        """vgx_reg_batch_alloc_outputs -> (residuals, jac_ref, jac_read device pointers (0 where not wanted), ms per launch)"""
        poses = _f64(poses).reshape(-1, 4)
        r, jo, je, ms = vp(), vp(), vp(), C.c_float()
        self.ctx.check(self.ctx.lib.vgx_reg_batch_alloc_outputs(self.h, _ptr(poses, f64p), poses.shape[0], int(n_candidates),
                                                                int(want_jac_ref), int(want_jac_read), C.byref(r), C.byref(jo),
                                                                C.byref(je), C.byref(ms)))
        return (r.value or 0), (jo.value or 0), (je.value or 0), float(ms.value)

    def free_outputs(self, r, jo, je):
        self.ctx.check(self.ctx.lib.vgx_reg_batch_free_outputs(self.h, vp(r) if r else None, vp(jo) if jo else None,
                                                               vp(je) if je else None))

    def evaluate_normal(self, poses, d_normal=None, to_host=True):
        poses = _f64(poses).reshape(-1, 4)
        status = np.zeros(max(self.n, 1), np.int32)
        host = np.zeros((self.n, NORMAL_SIZE), np.float64) if to_host else None
        self.ctx.check(self.ctx.lib.vgx_reg_batch_evaluate_normal(
            self.h, _ptr(poses, f64p), poses.shape[0], vp(d_normal) if d_normal else None,
            _ptr(host, f64p), _ptr(status, i32p)))
        return status[:self.n], host

    def evaluate_cost(self, poses, d_cost=None, to_host=True):
        """cost-only fused pass (vgx_reg_batch_evaluate_cost): -> (status, cost[n] f64 or None)"""
        poses = _f64(poses).reshape(-1, 4)
        status = np.zeros(max(self.n, 1), np.int32)
        host = np.zeros(self.n, np.float64) if to_host else None
        self.ctx.check(self.ctx.lib.vgx_reg_batch_evaluate_cost(
            self.h, _ptr(poses, f64p), poses.shape[0], vp(d_cost) if d_cost else None,
            _ptr(host, f64p), _ptr(status, i32p)))
        return status[:self.n], host

    def count_live(self, poses, unique=False):
        """residuals whose points the fused pass reads at these poses (chunk culling applied);
        with unique=True also the number of distinct points behind them"""
        poses = _f64(poses).reshape(-1, 4)
        n, u = C.c_int64(), C.c_int64()
        self.ctx.check(self.ctx.lib.vgx_reg_batch_count_live(self.h, _ptr(poses, f64p), poses.shape[0],
                                                             C.byref(n), C.byref(u) if unique else None))
        return (n.value, u.value) if unique else n.value

    def count_live_each(self, poses):
        """count_live per constraint (batch order)"""
        poses = _f64(poses).reshape(-1, 4)
        out = np.zeros(max(self.n, 1), np.int64)
        self.ctx.check(self.ctx.lib.vgx_reg_batch_count_live_each(self.h, _ptr(poses, f64p), poses.shape[0],
                                                                  out.ctypes.data_as(i64p)))
        return out[:self.n]

    def launch_order(self, points_pass):
        """1: constraints sharing a reference submap run side by side on one XCD (points shared in its
        L2); 0: constraint-major; -1: that pass has not run yet (vgx_reg_batch_launch_order)"""
        g = C.c_int32()
        self.ctx.check(self.ctx.lib.vgx_reg_batch_launch_order(self.h, int(bool(points_pass)), C.byref(g)))
        return g.value

    def assemble(self, n_nodes, d_fused, d_normal=None, zero_first=True):
        self.ctx.check(self.ctx.lib.vgx_reg_batch_assemble(
            self.h, vp(d_normal) if d_normal else None, n_nodes, vp(d_fused), int(zero_first)))

    def brick_layout(self):
        """BRICKS_APRON / BRICKS_QUAD: the bricks this batch reads (quad on demand when all its constraints sample)"""
        return int(self.ctx.lib.vgx_reg_batch_brick_layout(self.h))

    def scatter_normal(self, d_normal_all, d_normal=None, zero_first=True):
        """this shard's [n][45] blocks into rows global_index[c] of the DEVICE [n_global][45] array"""
        self.ctx.check(self.ctx.lib.vgx_reg_batch_scatter_normal(
            self.h, vp(d_normal) if d_normal else None, vp(d_normal_all), int(zero_first)))

    def destroy(self):
        if self.h:
            self.ctx.lib.vgx_reg_batch_destroy(self.h)
            self.h = None


class RegistrationAssembler:
    """vgx_reg_assembler: the whole constraint list's node structure on one context; builds the fused buffer from
    the complete [n][45] array in list order (the sharding-independent assembly, include/voxgraph_amd.h)."""

    def __init__(self, ctx, node_pair):
        self.ctx = ctx
        np_pair = np.ascontiguousarray(node_pair, dtype=np.int32).reshape(-1, 2)
        self.n = len(np_pair)
        h = vp()
        ctx.check(ctx.lib.vgx_reg_assembler_create(ctx.h, self.n, _ptr(np_pair, i32p), C.byref(h)))
        self.h = h

    def assemble(self, d_normal_all, n_nodes, d_fused):
        self.ctx.check(self.ctx.lib.vgx_reg_assembler_assemble(self.h, vp(d_normal_all) if d_normal_all else None,
                                                               n_nodes, vp(d_fused)))

    def destroy(self):
        if self.h:
            self.ctx.lib.vgx_reg_assembler_destroy(self.h)
            self.h = None


def lpt_shards(weights, n_shards):
    """vgx_lpt_shards: shard index per constraint (greedy longest-processing-time)."""
    w = np.ascontiguousarray(weights, np.int64)
    out = np.zeros(len(w), np.int32)
    rc = load().vgx_lpt_shards(len(w), _ptr(w, i64p), int(n_shards), _ptr(out, i32p))
    if rc != OK:
        raise VgxError(rc, "vgx_lpt_shards")
    return out


def contiguous_shards(weights, n_shards):
    """vgx_contiguous_shards: the list cut into n_shards consecutive runs of (nearly) equal weight"""
    w = np.ascontiguousarray(weights, np.int64)
    out = np.zeros(len(w), np.int32)
    rc = load().vgx_contiguous_shards(len(w), _ptr(w, i64p), int(n_shards), _ptr(out, i32p))
    if rc != OK:
        raise VgxError(rc, "vgx_contiguous_shards")
    return out


class RegistrationMulti:
    """vgx_reg_multi: the constraint list sharded over several contexts of this process."""

    def __init__(self, ctxs, cost_functions, node_pair):
        self.ctxs, self.n = list(ctxs), len(cost_functions)
        carr = (vp * len(self.ctxs))(*[c.h for c in self.ctxs])
        rarr = (vp * max(self.n, 1))(*[cf.h for cf in cost_functions])
        np_pair = np.ascontiguousarray(node_pair, dtype=np.int32).reshape(-1, 2)
        h = vp()
        self.ctxs[0].check(self.ctxs[0].lib.vgx_reg_multi_create(len(self.ctxs), carr, self.n, rarr,
                                                                 _ptr(np_pair, i32p), C.byref(h)))
        self.h = h
        self._keep = list(cost_functions)

    def shard_of(self):
        out = np.zeros(max(self.n, 1), np.int32)
        self.ctxs[0].check(self.ctxs[0].lib.vgx_reg_multi_shard_of(self.h, _ptr(out, i32p)))
        return out[:self.n]

    def evaluate_fused(self, poses):
        poses = _f64(poses).reshape(-1, 4)
        out = np.zeros(fused_size(poses.shape[0], self.n))
        status = np.zeros(max(self.n, 1), np.int32)
        self.ctxs[0].check(self.ctxs[0].lib.vgx_reg_multi_evaluate_fused(
            self.h, _ptr(poses, f64p), poses.shape[0], _ptr(out, f64p), _ptr(status, i32p)))
        return out, status[:self.n]

    def set_reduction(self, rccl):
        """False: the blocks gathered over peer mappings (default); True: one ncclAllReduce per evaluation"""
        self.ctxs[0].check(self.ctxs[0].lib.vgx_reg_multi_set_reduction(self.h, 1 if rccl else 0))

    def evaluate_normal(self, poses):
        poses = _f64(poses).reshape(-1, 4)
        out = np.zeros((self.n, NORMAL_SIZE))
        status = np.zeros(max(self.n, 1), np.int32)
        self.ctxs[0].check(self.ctxs[0].lib.vgx_reg_multi_evaluate_normal(
            self.h, _ptr(poses, f64p), poses.shape[0], _ptr(out, f64p), _ptr(status, i32p)))
        return out, status[:self.n]

    def evaluate_cost(self, poses):
        poses = _f64(poses).reshape(-1, 4)
        out = np.zeros(self.n)
        status = np.zeros(max(self.n, 1), np.int32)
        self.ctxs[0].check(self.ctxs[0].lib.vgx_reg_multi_evaluate_cost(
            self.h, _ptr(poses, f64p), poses.shape[0], _ptr(out, f64p), _ptr(status, i32p)))
        return out, status[:self.n]

    def destroy(self):
        if self.h:
            self.ctxs[0].lib.vgx_reg_multi_destroy(self.h)
            self.h = None


def atomic_roundtrip_ns(ctx, table_bytes=8 << 20, waves=1, chain=2000):
    """bench tooling (vgx_bench_atomic_roundtrip): ns per step of a chain of dependent device-scope exchanges"""
    out = C.c_float()
    ctx.check(ctx.lib.vgx_bench_atomic_roundtrip(ctx.h, table_bytes, waves, chain, C.byref(out)))
    return out.value


def alloc_scattered(ctx, nbytes, chunk_bytes, seed):
    """bench tooling (vgx_bench_alloc_scattered): device pointer of a range whose physical chunks are mapped in shuffled order"""
    out = vp()
    ctx.check(ctx.lib.vgx_bench_alloc_scattered(ctx.h, int(nbytes), int(chunk_bytes), int(seed), C.byref(out)))
    return int(out.value)


def free_scattered(ctx, ptr):
    ctx.check(ctx.lib.vgx_bench_free_scattered(ctx.h, vp(int(ptr))))


def stream_ceiling_ms(ctx, d_src, read_bytes, d_dst, write_bytes, launches=5):
    """bench tooling (vgx_bench_stream_ceiling): ms per launch that streams read_bytes in and write_bytes out"""
    out = C.c_float()
    ctx.check(ctx.lib.vgx_bench_stream_ceiling(ctx.h, d_src, int(read_bytes) & ~15, d_dst, int(write_bytes) & ~15,
                                               launches, C.byref(out)))
    return out.value


def synth_city_scan(ctx, sensor_pose, n_az, n_el, el_span, max_range, seed, d_points):
    """Benchmark tooling: sphere-traced LiDAR scan of the analytic city (sensor frame)."""
    ctx.check(ctx.lib.vgx_synth_city_scan(ctx.h, _ptr(_f64(sensor_pose), f64p), n_az, n_el,
                                          float(el_span), float(max_range), seed, vp(d_points)))


def find_overlapping_pairs(ctx, submaps, poses, max_pairs=None):
    """PoseGraphInterface::updateOverlappingSubmapList -> [(i, j)] with i < j."""
    n = len(submaps)
    arr = (vp * max(n, 1))(*[s.h for s in submaps])
    poses = _f64(poses).reshape(-1, 4)
    max_pairs = n * (n - 1) // 2 if max_pairs is None else max_pairs
    pairs = np.zeros((max(max_pairs, 1), 2), np.int32)
    k = C.c_int32()
    ctx.check(ctx.lib.vgx_find_overlapping_pairs(ctx.h, n, arr, _ptr(poses, f64p), _ptr(pairs, i32p),
                                                 max_pairs, C.byref(k)))
    return [tuple(int(x) for x in p) for p in pairs[:k.value]]


def compress_normal(normal45):
    """45-number normal block -> (r_c[9], J_c[9,8]) with identical normal equations."""
    nb = _f64(normal45)
    r, J = np.zeros(9), np.zeros((9, 8))
    rc = load().vgx_reg_compress_normal(_ptr(nb, f64p), _ptr(r, f64p), _ptr(J, f64p))
    if rc != OK:
        raise VgxError(rc, "vgx_reg_compress_normal")
    return r, J


def fused_size(n_nodes, n_global):
    return load().vgx_reg_fused_size(n_nodes, n_global)


# ----------------------------------------------------------------------------
# TSDF path
# ----------------------------------------------------------------------------
def esdf_config(**kw):
    cfg = EsdfConfig()
    load().vgx_esdf_config_default(C.byref(cfg))
    for k, v in kw.items():
        if not hasattr(cfg, k):
            raise AttributeError(k)
        setattr(cfg, k, v)
    return cfg


def tsdf_config(**kw):
    cfg = TsdfConfig()
    load().vgx_tsdf_config_default(C.byref(cfg))
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
    """voxblox::Layer<TsdfVoxel> of the active submap, resident on the GPU."""

    def __init__(self, ctx, voxel_size, vps, lut_min=None, lut_dim=None, max_blocks=0):
        """lut_min / lut_dim / max_blocks are only an initial reservation: the layer grows."""
        self.ctx, self.vps = ctx, vps
        mn = None if lut_min is None else np.ascontiguousarray(lut_min, np.int32)
        dm = None if lut_dim is None else np.ascontiguousarray(lut_dim, np.int32)
        h = vp()
        ctx.check(ctx.lib.vgx_tsdf_layer_create(ctx.h, float(voxel_size), vps, _ptr(mn, i32p),
                                                _ptr(dm, i32p), max_blocks, C.byref(h)))
        self.h = h

    def growths(self):
        return int(self.ctx.lib.vgx_tsdf_layer_growths(self.h))

    def clear_dropped(self):
        self.ctx.check(self.ctx.lib.vgx_tsdf_layer_clear_dropped(self.h))

    def reserve(self, origin, reach_m):
        o = _f32(origin)
        self.ctx.check(self.ctx.lib.vgx_tsdf_layer_reserve(self.h, _ptr(o, f32p), float(reach_m)))

    def upload(self, block_index, distance, weight, rgba=None):
        bi = np.ascontiguousarray(block_index, np.int32).reshape(-1, 3)
        d, w = _f32(distance), _f32(weight)
        c = None if rgba is None else np.ascontiguousarray(rgba, np.uint8)
        self.ctx.check(self.ctx.lib.vgx_tsdf_layer_upload(self.h, bi.shape[0], _ptr(bi, i32p), _ptr(d, f32p),
                                                          _ptr(w, f32p), _ptr(c, u8p)))

    def stats(self):
        n, d = C.c_int32(), C.c_int64()
        self.ctx.check(self.ctx.lib.vgx_tsdf_layer_stats(self.h, C.byref(n), C.byref(d)))
        return n.value, d.value

    def download(self):
        n, _ = self.stats()
        nv = self.vps ** 3
        bi = np.zeros((n, 3), np.int32)
        d = np.zeros((n, nv), np.float32)
        w = np.zeros((n, nv), np.float32)
        rgba = np.zeros((n, nv, 4), np.uint8)
        self.ctx.check(self.ctx.lib.vgx_tsdf_layer_download(self.h, _ptr(bi, i32p), _ptr(d, f32p),
                                                            _ptr(w, f32p), _ptr(rgba, u8p)))
        return bi, d, w, rgba

    def destroy(self):
        if self.h:
            self.ctx.lib.vgx_tsdf_layer_destroy(self.h)
            self.h = None


class FastTsdfIntegrator:
    """Mirror of voxblox::FastTsdfIntegrator as voxgraph drives it
    (pointcloud_integrator.cpp:66-83): ctor(config, layer), setLayer, integratePointCloud."""

    def __init__(self, ctx, config, layer):
        self.ctx, self.config, self.layer = ctx, config, layer
        h = vp()
        ctx.check(ctx.lib.vgx_tsdf_integrator_create(ctx.h, C.byref(config), layer.h, C.byref(h)))
        self.h = h

    def setLayer(self, layer):
        self.ctx.check(self.ctx.lib.vgx_tsdf_integrator_set_layer(self.h, layer.h))
        self.layer = layer

    def integratePointCloud(self, T_G_C, points_C, colors=None, freespace_points=False, count=True):
        """count=False: n_updates == NULL, voxblox's void call -- uncounted, returns with the scan queued"""
        T = _f32(T_G_C)
        pts = _f32(points_C).reshape(-1, 3)
        col = None if colors is None else np.ascontiguousarray(colors, np.uint8).reshape(-1, 4)
        n = C.c_int64()
        self.ctx.check(self.ctx.lib.vgx_tsdf_integrate(self.h, _ptr(T, f32p), _ptr(pts, f32p),
                                                       _ptr(col, u8p), pts.shape[0],
                                                       int(freespace_points), C.byref(n) if count else None))
        return n.value

    def integratePointCloudMerged(self, T_G_C, points_C, colors=None, freespace_points=False):
        """voxblox::MergedTsdfIntegrator::integratePointCloud (same config / layer)"""
        T = _f32(T_G_C)
        pts = _f32(points_C).reshape(-1, 3)
        col = None if colors is None else np.ascontiguousarray(colors, np.uint8).reshape(-1, 4)
        n = C.c_int64()
        self.ctx.check(self.ctx.lib.vgx_tsdf_integrate_merged(self.h, _ptr(T, f32p), _ptr(pts, f32p),
                                                              _ptr(col, u8p), pts.shape[0],
                                                              int(freespace_points), C.byref(n)))
        return n.value

    def integrate_merged_device(self, T_G_C, d_points, d_rgba, n, freespace_points=False, count=False):
        T = _f32(T_G_C)
        out = C.c_int64()
        self.ctx.check(self.ctx.lib.vgx_tsdf_integrate_merged_device(
            self.h, _ptr(T, f32p), vp(d_points), vp(d_rgba) if d_rgba else None, n,
            int(freespace_points), C.byref(out) if count else None))
        return out.value

    def integrate_device(self, T_G_C, d_points, d_rgba, n, freespace_points=False, count=False):
        T = _f32(T_G_C)
        out = C.c_int64()
        self.ctx.check(self.ctx.lib.vgx_tsdf_integrate_device(
            self.h, _ptr(T, f32p), vp(d_points), vp(d_rgba) if d_rgba else None, n,
            int(freespace_points), C.byref(out) if count else None))
        return out.value

    def set_cloud_width(self, width):
        """organised clouds: points per row (sensor_msgs/PointCloud2.width); 0 = unorganised"""
        self.ctx.check(self.ctx.lib.vgx_tsdf_integrator_set_cloud_width(self.h, int(width)))

    def set_speculation(self, depth=32, threshold=8 << 20):
        """test tooling (reproducible mode): write rays out `depth` steps deep at first when a scan's complete
        walks exceed `threshold` steps; the layer does not depend on either"""
        self.ctx.check(self.ctx.lib.vgx_tsdf_integrator_set_speculation(self.h, depth, threshold))

    def read_trace(self, max_workgroups):
        """last counted racing scan -> [workgroups][16] float64: four stamps in microseconds relative to the first start,
        then rays, rounds, per-voxel folds, longest chain of repeated folds, and the workgroup's eight statistics"""
        buf = np.zeros((int(max_workgroups), 16), np.int64)
        n, khz = C.c_int64(), C.c_int64()
        self.ctx.check(self.ctx.lib.vgx_tsdf_integrator_read_trace(self.h, _ptr(buf, i64p), int(max_workgroups),
                                                                   C.byref(n), C.byref(khz)))
        t = buf[:min(n.value, int(max_workgroups))].astype(np.float64)
        if len(t):
            t0 = t[:, 0].min()
            t[:, :4] = (t[:, :4] - t0) * 1e3 / max(khz.value, 1)
        return t

    def walk_stats(self):
        """bench tooling, last counted racing scan (include/voxgraph_amd_bench.h): dict of the seven numbers"""
        out = (C.c_int64 * 7)()
        self.ctx.check(self.ctx.lib.vgx_tsdf_integrator_walk_stats(self.h, out))
        names = ("longest_chain", "exchanges", "colour_blends", "peeks", "voxel_folds", "cas_retries", "overrun_exchanges")
        return {k: int(v) for k, v in zip(names, out)}

    def set_event_trace(self, capacity_words):
        """test tooling: racing scans from now on run the event-logging form of the shipped kernel (0: off)"""
        self.ctx.check(self.ctx.lib.vgx_tsdf_integrator_set_event_trace(self.h, int(capacity_words)))
        self._trace_cap = int(capacity_words)

    def read_event_trace(self):
        """-> (uint64 words of the log since the last read, events lost to a full log); empties the log"""
        buf = np.zeros(self._trace_cap, np.uint64)
        n, lost = C.c_int64(), C.c_int64()
        self.ctx.check(self.ctx.lib.vgx_tsdf_integrator_read_event_trace(
            self.h, _ptr(buf, C.POINTER(C.c_uint64)), len(buf), C.byref(n), C.byref(lost)))
        return buf[:n.value].copy(), int(lost.value)

    def download_sets(self):
        """-> (start set, observed set: uint64[2^20] each, (start offset, observed offset, scans since the last reset))"""
        a, b = np.zeros(1 << 20, np.uint64), np.zeros(1 << 20, np.uint64)
        st = (C.c_int64 * 3)()
        self.ctx.check(self.ctx.lib.vgx_tsdf_integrator_download_sets(
            self.h, _ptr(a, C.POINTER(C.c_uint64)), _ptr(b, C.POINTER(C.c_uint64)), st))
        return a, b, tuple(int(x) for x in st)

    def destroy(self):
        if self.h:
            self.ctx.lib.vgx_tsdf_integrator_destroy(self.h)
            self.h = None


# ------------------------------------------------------------------ saved maps
FILE_CBLOX_COLLECTION, FILE_VOXBLOX_LAYER = 0, 1


class MapFile:
    """A cblox submap-collection file (voxgraph's save_to_file) or a voxblox layer file.
    Host-only except load_submap()."""

    def __init__(self, path, fmt=FILE_CBLOX_COLLECTION):
        self.lib = load()
        h = vp()
        rc = self.lib.vgx_map_file_open(os.fsencode(path), fmt, C.byref(h))
        if rc != 0:
            raise VgxError(rc, self.lib.vgx_map_file_last_error(None).decode())
        self.h = h

    def _check(self, rc):
        if rc != 0:
            raise VgxError(rc, self.lib.vgx_map_file_last_error(self.h).decode())

    def __len__(self):
        return int(self.lib.vgx_map_file_num_submaps(self.h))

    def info(self, i):
        info = MapFileSubmapInfo()
        self._check(self.lib.vgx_map_file_get_submap_info(self.h, i, C.byref(info)))
        return info

    def read_submap(self, i, want_rgba=False):
        """-> dict(block_index, tsdf_distance, tsdf_weight, [tsdf_rgba], esdf_distance, esdf_observed)"""
        info = self.info(i)
        nb, vox = info.n_tsdf_blocks, info.voxels_per_side ** 3
        out = dict(block_index=np.zeros((nb, 3), np.int32), tsdf_distance=np.zeros((nb, vox), np.float32),
                   tsdf_weight=np.zeros((nb, vox), np.float32),
                   tsdf_rgba=np.zeros((nb, vox, 4), np.uint8) if want_rgba else None,
                   esdf_distance=np.zeros((nb, vox), np.float32), esdf_observed=np.zeros((nb, vox), np.uint8))
        self._check(self.lib.vgx_map_file_read_submap(
            self.h, i, _ptr(out["block_index"], i32p), _ptr(out["tsdf_distance"], f32p),
            _ptr(out["tsdf_weight"], f32p), _ptr(out["tsdf_rgba"], u8p),
            _ptr(out["esdf_distance"], f32p), _ptr(out["esdf_observed"], u8p)))
        return out

    def load_submap(self, ctx, i):
        """VoxgraphSubmap::LoadFromStream onto the device (not yet finished)."""
        sm = Submap.__new__(Submap)
        sm.ctx = ctx
        h = vp()
        self._check(self.lib.vgx_map_file_load_submap(ctx.h, self.h, i, C.byref(h)))
        sm.h = h
        return sm

    def close(self):
        if getattr(self, "h", None):
            self.lib.vgx_map_file_close(self.h)
            self.h = None

    __del__ = close


def write_map_file(path, fmt, voxel_size, vps, submaps):
    """submaps: list of dict(id, T_M_S[7], block_index, tsdf_distance, tsdf_weight,
    tsdf_rgba=None, esdf_distance=None, esdf_observed=None)"""
    lib = load()
    arr = (MapFileSubmapData * len(submaps))()
    keep = []
    for k, s in enumerate(submaps):
        bi = np.ascontiguousarray(s["block_index"], np.int32).reshape(-1, 3)
        td, tw = _f32(s["tsdf_distance"]), _f32(s["tsdf_weight"])
        rgba = None if s.get("tsdf_rgba") is None else np.ascontiguousarray(s["tsdf_rgba"], np.uint8)
        ed = _f32(s.get("esdf_distance"))
        eo = None if s.get("esdf_observed") is None else np.ascontiguousarray(s["esdf_observed"], np.uint8)
        keep += [bi, td, tw, rgba, ed, eo]
        arr[k].id = int(s.get("id", k))
        for a, v in enumerate(s.get("T_M_S", (1, 0, 0, 0, 0, 0, 0))):
            arr[k].T_M_S[a] = float(v)
        arr[k].n_blocks = bi.shape[0]
        arr[k].block_index, arr[k].tsdf_distance, arr[k].tsdf_weight = _ptr(bi, i32p), _ptr(td, f32p), _ptr(tw, f32p)
        arr[k].tsdf_rgba, arr[k].esdf_distance, arr[k].esdf_observed = _ptr(rgba, u8p), _ptr(ed, f32p), _ptr(eo, u8p)
    rc = lib.vgx_map_file_write(os.fsencode(path), fmt, float(voxel_size), int(vps), len(submaps), arr)
    if rc != 0:
        raise VgxError(rc, lib.vgx_map_file_last_error(None).decode())
