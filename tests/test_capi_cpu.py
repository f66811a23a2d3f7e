"""CPU-side checks of the drop-in boundary: the C-ABI library loads, exports every
symbol include/voxgraph_amd.h declares, and refuses to run without a GPU (no
fallback).  No compute calls here."""
import os
import re
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def capi():
    import __graft_entry__ as g
    g.build()
    from voxgraph_amd import capi
    return capi


def _declared_symbols(headers=("voxgraph_amd.h", "voxgraph_amd_bench.h")):
    text = "".join(open(os.path.join(ROOT, "include", h)).read() for h in headers)
    return sorted(set(re.findall(r"VGX_API\s+[\w\s\*]+?\b(vgx_\w+)\s*\(", text)))


def test_library_exports_every_declared_symbol(capi):
    lib = capi.load()
    boundary = _declared_symbols(("voxgraph_amd.h",))
    tooling = _declared_symbols(("voxgraph_amd_bench.h",))
    assert len(boundary) >= 30 and not set(boundary) & set(tooling)
    for name in boundary + tooling:
        assert hasattr(lib, name), f"{name} declared in include/ but not exported"
    # the ctypes tables cover the two headers exactly
    assert sorted(capi.SIGNATURES) == boundary
    assert sorted(capi.BENCH_SIGNATURES) == tooling
    # benchmark / test tooling lives in its own header AND its own library: the product library carries none of it
    assert not [n for n in boundary if "synth" in n or "bench" in n]
    product = _exported(capi.LIB_PATH)
    assert not set(tooling) & set(product), sorted(set(tooling) & set(product))
    assert set(tooling) <= set(_exported(capi.BENCH_LIB_PATH))
    assert tooling == ["vgx_bench_alloc_scattered", "vgx_bench_atomic_roundtrip", "vgx_bench_free_scattered", "vgx_bench_stream_ceiling",
                       "vgx_synth_city_scan", "vgx_synth_city_submap", "vgx_tsdf_integrator_download_sets",
                       "vgx_tsdf_integrator_read_event_trace", "vgx_tsdf_integrator_read_trace", "vgx_tsdf_integrator_set_event_trace",
                       "vgx_tsdf_integrator_set_speculation", "vgx_tsdf_integrator_walk_stats"]


def _exported(path):
    out = subprocess.check_output(["nm", "-D", "--defined-only", path]).decode()
    return [l.split()[-1] for l in out.splitlines() if " T " in l]


def test_only_c_abi_symbols_are_exported(capi):
    """the product library exports the boundary header's symbols and three C-linkage hooks for the tooling library
    (csrc/vgx_internal.h), nothing else -- no C++ names; the tooling library exports its header's symbols only"""
    names = _exported(capi.LIB_PATH)
    boundary = _declared_symbols(("voxgraph_amd.h",))
    hooks = ["vgx_internal_build_block_lut", "vgx_internal_launch_brickify", "vgx_internal_set_error"]
    assert sorted(names) == sorted(boundary + hooks), sorted(set(names) ^ set(boundary + hooks))
    assert sorted(_exported(capi.BENCH_LIB_PATH)) == _declared_symbols(("voxgraph_amd_bench.h",))


def test_product_does_not_link_or_reference_the_oracle(capi):
    out = subprocess.check_output(["ldd", capi.LIB_PATH]).decode()
    assert "oracle" not in out
    for dirpath, _, files in os.walk(os.path.join(ROOT, "voxgraph_amd")):
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".cpp")):
                src = open(os.path.join(dirpath, f), errors="ignore").read()
                for token in ("import oracle", "from oracle", "liboracle", "orc_", "pyoracle",
                              "reg_oracle.h", "ref_shims", "libref_reg", "refreg_", "ref_reg"):
                    assert token not in src, (dirpath, f, token)


def test_no_gpu_means_loud_failure_not_fallback(capi):
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    with pytest.raises(capi.VgxError) as e:
        capi.Context(0)
    assert e.value.code == capi.ERR_NO_DEVICE
    assert "no CPU fallback" in str(e.value)


def test_config_defaults_match_reference(capi):
    """registration_cost_function.h:20-35"""
    cfg = capi.default_config()
    assert cfg.registration_point_type == capi.POINTS_ISOSURFACE == 0
    assert cfg.sampling_ratio == -1
    assert cfg.no_correspondence_cost == 0
    assert cfg.use_esdf_distance == 1
    assert cfg.sampler_seed == 0          # the submap's shared sampler stream, as WeightedSampler
    assert capi.fused_size(200, 1000) == 1 + 20 * 200 + 16 * 1000


def test_header_is_plain_c(tmp_path):
    """the boundary is a C ABI: include/voxgraph_amd.h must compile as C99 (no C++ism, no torch /
    Ceres / Eigen types) and as C++"""
    src = tmp_path / "hdr.c"
    src.write_text('#include "voxgraph_amd.h"\n#include "voxgraph_amd_bench.h"\nint main(void) { vgx_reg_config c; vgx_tsdf_config t; '
                   'vgx_map_file_submap_info i; (void)c; (void)t; (void)i; return 0; }\n')
    inc = os.path.join(ROOT, "include")
    subprocess.check_call(["gcc", "-std=c99", "-pedantic", "-Wall", "-Wextra", "-Werror", "-I", inc,
                           "-fsyntax-only", str(src)])
    subprocess.check_call(["g++", "-std=c++14", "-Wall", "-Wextra", "-Werror", "-I", inc, "-x", "c++",
                           "-fsyntax-only", str(src)])


def test_lpt_shards_is_the_greedy_longest_processing_time_partition(capi):
    """vgx_lpt_shards (pure host arithmetic): the placement bench.py, the in-process multi-GPU
    component and a maintainer's own code share.  Against a direct restatement of the rule."""
    import numpy as np
    rng = np.random.default_rng(0)
    for n, k in ((0, 3), (1, 4), (50, 8), (1176, 8), (7, 1)):
        w = rng.integers(1, 500000, n)
        got = capi.lpt_shards(w, k)
        order = np.argsort(-w, kind="stable")
        load, want = np.zeros(k, np.int64), np.zeros(n, np.int32)
        for c in order:
            r = int(np.argmin(load))
            want[c] = r
            load[r] += w[c]
        assert np.array_equal(got, want)
        if n >= k:
            assert load.max() - load.min() <= w.max()


def test_contiguous_shards_cut_the_list_into_consecutive_runs_of_equal_weight(capi):
    """vgx_contiguous_shards (pure host arithmetic): the locality-aware placement -- monotone in the list order,
    every shard's weight within one constraint of the mean, and on the bench's config-3 graph a shard touches
    a third of the submaps where LPT touches three quarters."""
    import types
    import numpy as np
    import bench
    rng = np.random.default_rng(1)
    for n, k in ((0, 3), (1, 4), (50, 8), (1176, 8), (7, 1), (5, 8)):
        w = rng.integers(1, 500000, n)
        got = capi.contiguous_shards(w, k)
        assert len(got) == n and (n == 0 or (got.min() >= 0 and got.max() < k))
        assert np.all(np.diff(got) >= 0)                                       # consecutive runs
        if n >= 4 * k:
            load = np.bincount(got, weights=w, minlength=k)
            assert np.abs(load - w.sum() / k).max() <= w.max()                  # within one constraint of the mean
    assert np.array_equal(capi.contiguous_shards(np.zeros(6, np.int64), 3), [0, 0, 1, 1, 2, 2])
    args = types.SimpleNamespace(grid=[20, 10], block_dims=[16, 16, 16], voxel_size=0.2, seed=2, pose_sigma=0.3, yaw_sigma=0.05)
    _, _, pairs = bench.build_graph(args)
    w = np.full(len(pairs), 1000, np.int64)

    def touched(shard_of):
        return [len({int(s) for c in np.flatnonzero(shard_of == r) for s in pairs[c]}) for r in range(8)]
    lpt, cont = touched(capi.lpt_shards(w, 8)), touched(capi.contiguous_shards(w, 8))
    assert max(cont) <= 0.4 * 200 and min(lpt) >= 0.6 * 200, (lpt, cont)
