"""REG on several GPUs of one process (vgx_reg_multi_*) exercised with TWO CONTEXTS ON
ONE DEVICE: every code path of the multi-GPU component runs (LPT placement, one batch and one host
thread per context, per-context assembly, event-ordered fixed-order sum on context 0) except the
xGMI peer mapping itself, which a one-GPU box cannot offer."""
import numpy as np
import pytest

from oracle import synth
from tests import helpers as H

pytestmark = pytest.mark.gpu
PAIRS = [(0, 1), (1, 0), (0, 2), (1, 3), (2, 3), (3, 0), (2, 1)]


@pytest.fixture(scope="module")
def capi():
    from voxgraph_amd import capi as m
    m.load()
    return m


@pytest.fixture(scope="module")
def world(capi):
    """the same four submaps uploaded to two contexts (replicated, as on two GPUs)"""
    sdf = synth.sphere_ground_sdf((1.6, 1.6, 1.2), 1.0, 0.35)
    poses_true = [(0, 0, 0, 0), (0.8, 0.1, 0.0, 0.1), (0.1, 0.9, 0.05, -0.15), (0.9, 0.8, 0.0, 0.2)]
    ctxs = [capi.Context(0), capi.Context(0)]
    subs = [[], []]
    n_pts = []
    for i, p in enumerate(poses_true):
        sm = synth.make_submap(sdf, 0.1, 16, (0, 0, 0), (2, 2, 2), trunc=0.3, esdf_max=1.0, pose=p,
                               drop_empty_blocks=True)
        for k in range(2):
            g = H.gpu_submap(capi, ctxs[k], sm, i)
            n = g.extract_voxel_points()
            subs[k].append(g)
        n_pts.append(n)
    rng = np.random.default_rng(3)
    poses = np.array(poses_true, np.float64) + rng.normal(0, 0.03, (4, 4))
    yield dict(ctxs=ctxs, subs=subs, n_pts=n_pts, poses=poses)
    for k in range(2):
        for g in subs[k]:
            g.destroy()
        ctxs[k].close()


def test_two_contexts_sum_to_the_unsharded_buffer(capi, world):
    ctxs, subs, poses = world["ctxs"], world["subs"], world["poses"]
    cfg = capi.default_config(registration_point_type=capi.POINTS_VOXELS)
    weights = [world["n_pts"][a] for a, _ in PAIRS]
    shard = capi.lpt_shards(weights, 2)
    assert set(shard) == {0, 1}
    load = [sum(w for w, s in zip(weights, shard) if s == k) for k in range(2)]
    assert abs(load[0] - load[1]) <= max(weights)                      # LPT guarantee
    cfs = [capi.RegistrationCostFunction(ctxs[shard[c]], subs[shard[c]][a], subs[shard[c]][b], cfg)
           for c, (a, b) in enumerate(PAIRS)]
    multi = capi.RegistrationMulti(ctxs, cfs, PAIRS)
    assert np.array_equal(multi.shard_of(), shard)
    fused, status = multi.evaluate_fused(poses)
    assert np.all(status == 0)
    # the unsharded reference: everything on context 0
    import torch
    cfs0 = [capi.RegistrationCostFunction(ctxs[0], subs[0][a], subs[0][b], cfg) for a, b in PAIRS]
    single = capi.RegistrationBatch(ctxs[0], cfs0, PAIRS)
    _, normal0 = single.evaluate_normal(poses)
    buf = torch.zeros(capi.fused_size(4, len(PAIRS)), dtype=torch.float64, device="cuda:0")
    torch.cuda.synchronize()
    single.assemble(4, buf.data_ptr(), zero_first=True)
    ctxs[0].synchronize()
    want = buf.cpu().numpy()
    assert want[0] > 0
    np.testing.assert_allclose(fused, want, rtol=1e-12, atol=1e-9 * np.abs(want).max())
    # per-constraint blocks come back in the caller's order and do not depend on the placement
    normal, status = multi.evaluate_normal(poses)
    assert np.all(status == 0)
    np.testing.assert_allclose(normal, normal0, rtol=1e-12, atol=1e-12 * np.abs(normal0).max())
    # bitwise reproducible, evaluation after evaluation (fixed-order reduction, no atomics)
    for _ in range(3):
        f2, _ = multi.evaluate_fused(poses)
        assert np.array_equal(f2, fused)
    # a different evaluation point changes the answer (nothing cached across calls)
    p2 = poses.copy()
    p2[1, 0] += 0.04
    f3, _ = multi.evaluate_fused(p2)
    assert abs(f3[0] - fused[0]) > 1e-6 * fused[0]
    # the harness solver driven by the multi-context backend lands where the single batch does
    from harness import lm
    from harness.backends import GpuBackend

    class MultiBackend:
        def __call__(self, x):
            return multi.evaluate_fused(x)[0]
    info = [1.0, 1.0, 2500.0, 2500.0]
    edges = [lm.RelativePoseEdge.from_poses(k, k + 1, poses[k], poses[k + 1], info) for k in range(3)]
    kw = dict(parameter_tolerance=1e-10, max_seconds=1e9)
    x_m, s_m = lm.solve(lm.Problem(MultiBackend(), 4, PAIRS, edges), poses, **kw)
    x_s, s_s = lm.solve(lm.Problem(GpuBackend(capi, ctxs[0], single, 4), 4, PAIRS, edges), poses, **kw)
    assert s_m["iterations"] == s_s["iterations"]
    assert np.abs(x_m - x_s).max() < 1e-9
    multi.destroy()
    single.destroy()
    for o in cfs + cfs0:
        o.destroy()


def test_a_context_without_constraints_and_foreign_constraints(capi, world):
    ctxs, subs, poses = world["ctxs"], world["subs"], world["poses"]
    cfg = capi.default_config(registration_point_type=capi.POINTS_VOXELS)
    import torch
    # dirty the device memory an empty shard's buffer is likely to be carved from (the
    # off-diagonal part of an empty shard's buffer used to be summed without ever being written):
    # a multi whose context 1 DOES own constraints writes non-zero blocks there, then frees them
    cfs1 = [capi.RegistrationCostFunction(ctxs[1], subs[1][a], subs[1][b], cfg) for a, b in PAIRS[:3]]
    dirty = capi.RegistrationMulti(ctxs, cfs1, PAIRS[:3])
    dirty.evaluate_fused(poses)
    dirty.destroy()
    cfs = [capi.RegistrationCostFunction(ctxs[0], subs[0][a], subs[0][b], cfg) for a, b in PAIRS[:3]]
    multi = capi.RegistrationMulti(ctxs, cfs, PAIRS[:3])          # context 1 gets nothing
    fused, _ = multi.evaluate_fused(poses)
    single = capi.RegistrationBatch(ctxs[0], cfs, PAIRS[:3])
    status, normal = single.evaluate_normal(poses)
    assert abs(fused[0] - normal[:, 0].sum()) <= 1e-12 * fused[0]
    # the WHOLE buffer, off-diagonal blocks included, equals the unsharded assembly
    buf = torch.full((capi.fused_size(4, 3),), float("nan"), dtype=torch.float64, device="cuda:0")
    torch.cuda.synchronize()
    single.assemble(4, buf.data_ptr(), zero_first=True)
    ctxs[0].synchronize()
    want = buf.cpu().numpy()
    np.testing.assert_allclose(fused, want, rtol=1e-12, atol=1e-9 * np.abs(want).max())
    # more contexts than constraints: one constraint, two contexts
    one = capi.RegistrationMulti(ctxs, cfs[:1], PAIRS[:1])
    f1, _ = one.evaluate_fused(poses)
    s1 = capi.RegistrationBatch(ctxs[0], cfs[:1], PAIRS[:1])
    buf1 = torch.full((capi.fused_size(4, 1),), float("nan"), dtype=torch.float64, device="cuda:0")
    torch.cuda.synchronize()
    s1.evaluate_normal(poses)
    s1.assemble(4, buf1.data_ptr(), zero_first=True)
    ctxs[0].synchronize()
    np.testing.assert_allclose(f1, buf1.cpu().numpy(), rtol=1e-12, atol=1e-9 * np.abs(want).max())
    one.destroy()
    s1.destroy()
    multi.destroy()
    for o in cfs1:
        o.destroy()
    single.destroy()
    third = capi.Context(0)
    with pytest.raises(capi.VgxError):
        capi.RegistrationMulti([third], cfs, PAIRS[:3])              # constraints of another context
    third.close()
    for o in cfs:
        o.destroy()


def test_rccl_reduction_variant(capi, world):
    """vgx_reg_multi_set_reduction(VGX_REDUCE_RCCL): one ncclAllReduce of the fused buffer per evaluation
    instead of the fixed-order peer sum (BASELINE north_star names an RCCL all-reduce of the stacked
    J^T r).  RCCL wants one device per rank, so on a one-GPU box the variant can only run with ONE
    context (communicator of size 1: the whole call sequence -- dlopen, ncclCommInitAll, grouped
    ncclAllReduce on the context's stream, copy back -- with a trivial reduction); two contexts on one
    device must be refused, not hang."""
    ctxs, subs, poses = world["ctxs"], world["subs"], world["poses"]
    cfg = capi.default_config(registration_point_type=capi.POINTS_VOXELS)
    cfs = [capi.RegistrationCostFunction(ctxs[0], subs[0][a], subs[0][b], cfg) for a, b in PAIRS]
    one = capi.RegistrationMulti(ctxs[:1], cfs, PAIRS)
    want, _ = one.evaluate_fused(poses)
    one.set_reduction(True)
    for _ in range(2):
        got, status = one.evaluate_fused(poses)
        assert np.all(status == 0) and np.array_equal(got, want)
    one.set_reduction(False)
    assert np.array_equal(one.evaluate_fused(poses)[0], want)
    one.destroy()
    cfs1 = [capi.RegistrationCostFunction(ctxs[1], subs[1][a], subs[1][b], cfg) for a, b in PAIRS[:2]]
    two = capi.RegistrationMulti(ctxs, cfs[:3] + cfs1, PAIRS[:3] + PAIRS[:2])
    with pytest.raises(capi.VgxError):
        two.set_reduction(True)                          # both contexts sit on device 0
    two.evaluate_fused(poses)                            # still usable with the default reduction
    two.destroy()
    for o in cfs + cfs1:
        o.destroy()
