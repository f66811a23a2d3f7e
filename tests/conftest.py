import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu)")
    # A checkout without the built library (the .so is git-ignored): build it once, loudly, with
    # hipcc -- the same __graft_entry__.build() the driver runs.  Never a CPU substitute.
    lib = os.path.join(ROOT, "voxgraph_amd", "lib", "libvoxgraph_amd.so")
    if not os.path.exists(lib):
        import __graft_entry__
        __graft_entry__.build()


def _gfx950_present():
    """True when a gfx950 device can be opened (asked of the product library, not of torch)."""
    try:
        from voxgraph_amd import capi
        capi.Context(0).close()
        return True
    except Exception:
        return False


# Execution order of the suite: `pytest -x` stops at the first failure, so the
# deterministic bit-exact parity tests -- the evidence every row of SURVEY.md 8 rests on -- run
# FIRST, the boundary / adapter / error-path tests next, and whatever is statistical or end-to-end
# (racing TSDF scans compared in distribution, whole sessions, the bench line) LAST.  Files not
# listed run in the middle, in their usual order; within a file the definition order is kept.
_ORDER = [
    # 1. REG hot path, bit-exact against the oracle / the reference's own source
    "test_reg_gpu", "test_fullsize_gpu", "test_batch_sampling_gpu", "test_dropin_cpp_gpu", "test_brick_layout_gpu",
    # 2. TSDF: the reproducible mode, the racing mode replayed event by event, the order-independent (bit-exact) cases
    "test_tsdf_deterministic_gpu", "test_tsdf_replay_gpu", "test_tsdf_gpu", "test_tsdf_merged_gpu", "test_tsdf_dropin_gpu",
    # 3. producers either side of the path, bit-exact
    "test_overlap_gpu", "test_iso_gpu", "test_esdf_gpu", "test_mapfile_gpu",
    # 4. boundary, adapters, sharding, error paths
    "test_multi_gpu", "test_cpp_adapter", "test_errors_gpu", "test_solve_gpu",
]
_LAST = ["test_pipeline_gpu", "test_chain_compare_gpu", "test_bench_gpu"]
# statistical tests inside otherwise deterministic files: moved behind every deterministic file
_STATISTICAL = {
    "test_full_scan_with_shipped_config_agrees_statistically",
    "test_merged_and_fast_agree_on_the_surface_they_reconstruct",
}


def _rank(item):
    stem = os.path.splitext(os.path.basename(str(item.fspath)))[0]
    name = item.name.split("[")[0]
    if stem in _LAST:
        return 3000 + _LAST.index(stem)
    if name in _STATISTICAL:
        return 2000
    if stem in _ORDER:
        return _ORDER.index(stem)
    return 1000


def pytest_collection_modifyitems(config, items):
    items.sort(key=_rank)            # stable: definition order survives inside a rank
    # plain `pytest` on a host without an MI355X: skip the gpu-marked tests instead of erroring
    # (`-m gpu` on a GPU box and `-m "not gpu"` here are unaffected)
    markexpr = (config.getoption("markexpr", "") or "").strip()
    if markexpr == "gpu":
        return          # the GPU tests were asked for by name: without a device they must fail loudly
    gpu_items = [it for it in items if "gpu" in it.keywords]
    if not gpu_items or _gfx950_present():
        return
    skip = pytest.mark.skip(reason="needs an MI355X (gfx950): vgx_ctx_create found no device")
    for it in gpu_items:
        it.add_marker(skip)


@pytest.fixture(scope="session")
def golden_dir():
    return os.path.join(ROOT, "tests", "golden")
