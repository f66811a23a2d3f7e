"""REG on several GPUs of one process (vgx_reg_multi_*) exercised with 2 ... 8 CONTEXTS ON
ONE DEVICE: every code path of the multi-GPU component runs (LPT placement, one batch and one host
thread per context, event-ordered gather of the per-constraint blocks on context 0, ONE assembly in list
order) except the xGMI peer mapping itself, which a one-GPU box cannot offer.  Round 4: the fused buffer
is the single batch's BIT FOR BIT, whatever the number of contexts and the placement."""
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
    assert np.array_equal(fused, want)             # the blocks are gathered, the assembly runs once, in list order
    # per-constraint blocks come back in the caller's order and do not depend on the placement
    normal, status = multi.evaluate_normal(poses)
    assert np.all(status == 0)
    assert np.array_equal(normal, normal0)         # a constraint's tile size is a function of the constraint alone
    # the cost-only pass on every context's share (vgx_reg_multi_evaluate_cost, what Ceres asks for at trial steps): element 0
    # of those blocks bit for bit, in the caller's order, and the single batch's cost-only pass says the same
    cost, status = multi.evaluate_cost(poses)
    assert np.all(status == 0)
    assert np.array_equal(cost.view(np.uint64), normal0[:, 0].copy().view(np.uint64))
    _, cost0 = single.evaluate_cost(poses)
    assert np.array_equal(cost.view(np.uint64), cost0.view(np.uint64))
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
    assert s_m["iterations"] == s_s["iterations"] and s_m["final_cost"] == s_s["final_cost"]
    assert np.array_equal(x_m, x_s)
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
    assert np.array_equal(fused, want)
    # more contexts than constraints: one constraint, two contexts
    one = capi.RegistrationMulti(ctxs, cfs[:1], PAIRS[:1])
    f1, _ = one.evaluate_fused(poses)
    s1 = capi.RegistrationBatch(ctxs[0], cfs[:1], PAIRS[:1])
    buf1 = torch.full((capi.fused_size(4, 1),), float("nan"), dtype=torch.float64, device="cuda:0")
    torch.cuda.synchronize()
    s1.evaluate_normal(poses)
    s1.assemble(4, buf1.data_ptr(), zero_first=True)
    ctxs[0].synchronize()
    assert np.array_equal(f1, buf1.cpu().numpy())
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


def _single_batch_fused(capi, ctx, subs0, pairs, cfg, poses, n_nodes):
    import torch
    cfs0 = [capi.RegistrationCostFunction(ctx, subs0[a], subs0[b], cfg) for a, b in pairs]
    single = capi.RegistrationBatch(ctx, cfs0, pairs)
    single.evaluate_normal(poses, to_host=False)
    buf = torch.full((capi.fused_size(n_nodes, len(pairs)),), float("nan"), dtype=torch.float64, device="cuda:0")
    torch.cuda.synchronize()
    single.assemble(n_nodes, buf.data_ptr(), zero_first=True)
    ctx.synchronize()
    want = buf.cpu().numpy()
    single.destroy()
    for o in cfs0:
        o.destroy()
    return want


def test_eight_contexts_small_graph_bit_identical_with_an_empty_shard(capi, world):
    """north_star's N = 8: EIGHT contexts on the one device, seven constraints -- so at
    least one context owns nothing -- and 3 / 5 contexts for good measure: the fused buffer and the
    per-constraint blocks are the single batch's BIT FOR BIT at every N, evaluation after evaluation."""
    ctxs2, subs2, poses = world["ctxs"], world["subs"], world["poses"]
    cfg = capi.default_config(registration_point_type=capi.POINTS_VOXELS)
    weights = [world["n_pts"][a] for a, _ in PAIRS]
    want = _single_batch_fused(capi, ctxs2[0], subs2[0], PAIRS, cfg, poses, 4)
    sdf = synth.sphere_ground_sdf((1.6, 1.6, 1.2), 1.0, 0.35)
    poses_true = [(0, 0, 0, 0), (0.8, 0.1, 0.0, 0.1), (0.1, 0.9, 0.05, -0.15), (0.9, 0.8, 0.0, 0.2)]
    extra = [capi.Context(0) for _ in range(6)]
    ctxs = list(ctxs2) + extra
    subs = [subs2[0], subs2[1]]
    for c in extra:                                           # replicate the four submaps on every context
        mine = []
        for i, p in enumerate(poses_true):
            sm = synth.make_submap(sdf, 0.1, 16, (0, 0, 0), (2, 2, 2), trunc=0.3, esdf_max=1.0, pose=p,
                                   drop_empty_blocks=True)
            g = H.gpu_submap(capi, c, sm, i)
            g.extract_voxel_points()
            mine.append(g)
        subs.append(mine)
    try:
        for n_ctx in (8, 5, 3):
            shard = capi.lpt_shards(weights, n_ctx)
            if n_ctx == 8:
                assert len(set(shard)) == 7                    # seven constraints on eight contexts: one is empty
            cfs = [capi.RegistrationCostFunction(ctxs[shard[c]], subs[shard[c]][a], subs[shard[c]][b], cfg)
                   for c, (a, b) in enumerate(PAIRS)]
            multi = capi.RegistrationMulti(ctxs[:n_ctx], cfs, PAIRS)
            assert np.array_equal(multi.shard_of(), shard)
            for _ in range(2):
                fused, status = multi.evaluate_fused(poses)
                assert np.all(status == 0) and np.array_equal(fused, want), (n_ctx, np.abs(fused - want).max())
            multi.destroy()
            for o in cfs:
                o.destroy()
    finally:
        for mine in subs[2:]:
            for g in mine:
                g.destroy()
        for c in extra:
            c.close()


def test_eight_contexts_on_a_64_submap_slice_of_config_3(capi):
    """the bench's own graph in small: 64 submaps of the config-3 city scene (8 x 8 grid, 128^3 voxels each so that
    eight replicas fit comfortably), ~350 overlap constraints, LPT-sharded by bytes moved over EIGHT contexts on the
    one device: the fused buffer equals the single batch's bit for bit; a shard that fails (a node index beyond
    n_nodes reaches only some shards' constraints) fails the whole evaluation with that shard's message."""
    import types
    import bench
    args = types.SimpleNamespace(grid=[8, 8], block_dims=[8, 8, 8], block_min=[-4, -4, -2], voxel_size=0.2,
                                 truncation=0.6, esdf_max=2.0, pose_sigma=0.3, yaw_sigma=0.05, seed=2)
    true_poses, poses, pairs = bench.build_graph(args)
    n_sub, n_con = len(true_poses), len(pairs)
    assert n_sub == 64 and n_con > 200
    ctxs = [capi.Context(0) for _ in range(8)]
    cfg = capi.default_config(registration_point_type=capi.POINTS_VOXELS)

    def city(ctx, k):
        sm = capi.Submap.synth_city(ctx, k, 0.2, 16, args.block_min, args.block_dims, 0.6, 2.0, 10.0, true_poses[k], 2)
        n = sm.extract_voxel_points(1.0, 0.3, True)
        sm.release_raw_layers()
        return sm, n
    subs0, n_points = zip(*[city(ctxs[0], k) for k in range(n_sub)])
    cfs0 = [capi.RegistrationCostFunction(ctxs[0], subs0[a], subs0[b], cfg) for a, b in pairs]
    probe = capi.RegistrationBatch(ctxs[0], cfs0, pairs)
    weights = 36 * np.array([n_points[a] for a, _ in pairs], np.int64) + 45 * probe.count_live_each(poses)
    probe.destroy()
    for o in cfs0:
        o.destroy()
    want = _single_batch_fused(capi, ctxs[0], subs0, pairs, cfg, poses, n_sub)
    shard = capi.lpt_shards(weights, 8)
    assert len(set(shard)) == 8
    load = np.bincount(shard, weights=weights, minlength=8)
    assert load.max() / load.mean() < 1.05                          # LPT on ~350 constraints: within 5 %
    subs = [dict(enumerate(subs0))] + [dict() for _ in range(7)]
    touched = []
    for k in range(1, 8):
        for s_ in sorted({int(x) for c in range(n_con) if shard[c] == k for x in pairs[c]}):
            subs[k][s_] = city(ctxs[k], s_)[0]
    for k in range(8):
        touched.append(len({int(x) for c in range(n_con) if shard[c] == k for x in pairs[c]}))
    print("constraints", n_con, "per context", np.bincount(shard).tolist(), "distinct submaps per context", touched)
    cfs = [capi.RegistrationCostFunction(ctxs[shard[c]], subs[shard[c]][int(a)], subs[shard[c]][int(b)], cfg)
           for c, (a, b) in enumerate(pairs)]
    multi = capi.RegistrationMulti(ctxs, cfs, pairs)
    for _ in range(2):
        fused, status = multi.evaluate_fused(poses)
        assert np.all(status == 0)
        assert np.array_equal(fused, want), np.abs(fused - want).max()
    normal, _ = multi.evaluate_normal(poses)
    assert abs(normal[:, 0].sum() - want[0]) <= 1e-12 * want[0]
    # a failing shard: with n_nodes = 40 only the shards holding a constraint on a node >= 40 fail; the call
    # reports that shard's error and the component stays usable afterwards
    with pytest.raises(capi.VgxError) as e:
        multi.evaluate_fused(poses[:40])
    assert "shard failed" in str(e.value) and "n_nodes" in str(e.value), str(e.value)
    fused2, _ = multi.evaluate_fused(poses)
    assert np.array_equal(fused2, want)
    multi.destroy()
    for o in cfs:
        o.destroy()
    for k in range(8):
        for sm in subs[k].values():
            sm.destroy()
        ctxs[k].close()
