"""The sort-based TSDF paths' own stable (key, position) sort (csrc/vgx_slot_sort.hip) against numpy's stable
argsort: the order voxblox's single-threaded integrators visit things in is what the reproducible mode
reproduces (pointcloud_integrator.cpp:83), so "equal keys keep their order" is part of the contract."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _sort(capi, ctx, torch, keys, end_bit, repeats=3):
    d_keys = torch.from_numpy(keys.view(np.int32)).cuda()
    d_out = torch.full((len(keys),), -1, dtype=torch.int32, device="cuda")
    d_idx = torch.full((len(keys),), -1, dtype=torch.int32, device="cuda")
    torch.cuda.synchronize()
    ms = capi.slot_sort(ctx, d_keys.data_ptr(), len(keys), end_bit, d_out.data_ptr(), d_idx.data_ptr(), repeats=repeats)
    ctx.synchronize()
    return d_out.cpu().numpy().view(np.uint32), d_idx.cpu().numpy().view(np.uint32), ms


def _check(capi, ctx, torch, keys, end_bit):
    got_keys, got_idx, _ = _sort(capi, ctx, torch, keys, end_bit)
    want_idx = np.argsort(keys, kind="stable").astype(np.uint32)
    assert np.array_equal(got_idx, want_idx), (len(keys), end_bit, int(np.flatnonzero(got_idx != want_idx)[0]))
    assert np.array_equal(got_keys, keys[want_idx])


@pytest.mark.parametrize("n", [4097, 5000, 8192, 8193, 65536, 100_003, 237_568, 1 << 20, 3_000_001])
@pytest.mark.parametrize("end_bit", [20, 21])
def test_random_slots_match_numpy_stable_argsort(n, end_bit):
    import torch
    from voxgraph_amd import capi
    capi.load()
    ctx = capi.Context(0)
    rng = np.random.default_rng(n + end_bit)
    _check(capi, ctx, torch, rng.integers(0, 1 << end_bit, n, dtype=np.uint32), end_bit)
    ctx.close()


@pytest.mark.parametrize("end_bit", [1, 8, 9, 16, 17, 24, 25, 32])
def test_every_number_of_passes(end_bit):
    import torch
    from voxgraph_amd import capi
    capi.load()
    ctx = capi.Context(0)
    rng = np.random.default_rng(end_bit)
    hi = (1 << end_bit) - 1
    keys = rng.integers(0, hi + 1, 150_001, dtype=np.uint64).astype(np.uint32)
    _check(capi, ctx, torch, keys, end_bit)
    ctx.close()


def test_runs_of_equal_keys_keep_their_order():
    """what the TSDF paths feed it: a few hot slots hit by thousands of records, most slots by one or two; plus the
    degenerate inputs (all equal, already sorted, reversed, one digit dominating)"""
    import torch
    from voxgraph_amd import capi
    capi.load()
    ctx = capi.Context(0)
    rng = np.random.default_rng(7)
    n = 300_000
    hot = rng.integers(0, 1 << 20, 12, dtype=np.uint32)
    keys = rng.integers(0, 1 << 20, n, dtype=np.uint32)
    pick = rng.random(n) < 0.4
    keys[pick] = hot[rng.integers(0, len(hot), int(pick.sum()))]
    _check(capi, ctx, torch, keys, 20)
    _check(capi, ctx, torch, np.full(50_000, 0x5a5a5, np.uint32), 20)
    _check(capi, ctx, torch, np.arange(70_000, dtype=np.uint32), 20)
    _check(capi, ctx, torch, np.arange(70_000, dtype=np.uint32)[::-1].copy(), 20)
    _check(capi, ctx, torch, (rng.integers(0, 16, 99_999, dtype=np.uint32) << 8) | 0x33, 20)
    _check(capi, ctx, torch, rng.integers(0, 3, 1_000_000, dtype=np.uint32), 21)
    ctx.close()


def test_timing_is_reported():
    import torch
    from voxgraph_amd import capi
    capi.load()
    ctx = capi.Context(0)
    keys = np.random.default_rng(1).integers(0, 1 << 20, 237_568, dtype=np.uint32)
    _, _, ms = _sort(capi, ctx, torch, keys, 20, repeats=20)
    assert 0.0 < ms < 5.0
    ctx.close()
