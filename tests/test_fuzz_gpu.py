"""A short run of each differential fuzzer (profiles/fuzz_*.py: random configurations / poses / clusters /
sampling set-ups against the oracles, exact comparisons) inside the suite, so that the driver's GPU run
exercises them too.  Fixed seeds: deterministic.  The long runs are recorded in profiles/README.md."""
import importlib.util
import os

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(script, seeds, first):
    spec = importlib.util.spec_from_file_location(script, os.path.join(ROOT, "profiles", script + ".py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    old = {k: os.environ.get(k) for k in ("SEEDS", "FIRST")}
    os.environ["SEEDS"], os.environ["FIRST"] = str(seeds), str(first)
    try:
        return mod.main()
    finally:
        for k, v in old.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v


def test_tsdf_random_configurations_bit_identical_to_the_oracle():
    """fast + merged integrators in the reproducible mode, merged in the racing mode: 25 random configurations"""
    assert _run("fuzz_tsdf", 25, 300) == 0


def test_racing_tsdf_random_configurations_replay_as_legal_interleavings():
    """the racing (default) mode by trace and replay: 12 random integrator configurations x 4 scans (unorganised and
    organised clouds incl. ragged tiles, counted and uncounted kernels, colours, free-space scans); the long runs
    (1360 scans, 59 M exchanges, 0 violations) are recorded in profiles/README.md"""
    assert _run("fuzz_replay", 12, 300) == 0


def test_reg_random_constraints_equal_the_oracle():
    """drop-in f64 rows, batched f32 rows, fused sums, voxel-point and isosurface producers: 60 random submap pairs"""
    assert _run("fuzz_reg", 60, 300) == 0


def test_overlap_random_clusters_equal_the_oracle():
    assert _run("fuzz_overlap", 40, 300) == 0


def test_sampling_random_setups_follow_the_oracles_streams():
    assert _run("fuzz_sampling", 30, 300) == 0


def test_reg_at_128_cubed_every_row_equal_and_fused_sums_close():
    """four 128^3 city submaps, twelve constraints, two pose sets per seed: every materialised row exact, fused
    sums within 2e-6 (this is the fuzzer that made the fused kernel share the reference's interpolated value)"""
    assert _run("fuzz_reg_large", 3, 300) == 0


def test_sampling_at_128_cubed_follows_the_oracles_streams():
    assert _run("fuzz_sampling_large", 3, 300) == 0


def test_sharded_evaluation_is_the_single_batch_bit_for_bit():
    """random pose graphs sharded over 1-8 contexts by LPT / contiguous / arbitrary placements: fused buffer,
    per-constraint blocks and the scatter -> int64 sum -> assemble route equal the single batch's bits (25 graphs)"""
    assert _run("fuzz_multi", 25, 700) == 0
