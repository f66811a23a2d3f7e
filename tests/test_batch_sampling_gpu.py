"""Batched evaluation of SAMPLING constraints -- the reference's shipped configuration
(voxgraph_mapper.yaml:34-35: sampling_ratio 0.05; registration_cost_function.cpp:113-122).

One evaluation of a batch is one Evaluate of every constraint in list order.  Constraints that
share a reference submap share its WeightedSampler engine (weighted_sampler.h:36-39), so they
consume consecutive ranges of ONE std::mt19937 stream; the batch generates those streams on the
device (mt_generate_kernel) and must reproduce, value for value, what the reference produces
when its cost functions are called in the same order."""
import numpy as np
import pytest

from oracle import pyoracle as orc
from oracle import ref_reg, synth
from tests import helpers as H
from tests.test_ref_pin import sequential_cumsum

pytestmark = pytest.mark.gpu
F = np.float32
RATIO = 0.3
# three constraints share reference submap 0, two share submap 1 (interleaved)
PAIRS = [(0, 1), (0, 2), (1, 0), (0, 3), (1, 3)]


@pytest.fixture(scope="module")
def capi():
    from voxgraph_amd import capi as m
    m.load()
    return m


@pytest.fixture(scope="module")
def ctx(capi):
    c = capi.Context(0)
    yield c
    c.close()


@pytest.fixture(scope="module")
def graph(capi, ctx):
    sdf = synth.sphere_ground_sdf((1.6, 1.6, 1.2), 1.0, 0.35)
    poses_true = [(0, 0, 0, 0), (0.8, 0.1, 0.0, 0.1), (0.1, 0.9, 0.05, -0.15), (0.9, 0.8, 0.0, 0.2)]
    rng = np.random.default_rng(5)
    sms, pts = [], []
    for p in poses_true:
        sm = synth.make_submap(sdf, 0.1, 16, (0, 0, 0), (2, 2, 2), trunc=0.3, esdf_max=1.0, pose=p,
                               drop_empty_blocks=True)
        xyz, dist, w = H.oracle_points(sm)
        w = (w * rng.uniform(0.2, 1.0, len(w))).astype(F)        # non-uniform weights: a real weighted draw
        sms.append(sm), pts.append((xyz, dist, w))
    poses = np.array(poses_true, np.float64) + rng.normal(0, 0.03, (4, 4))
    return dict(sms=sms, pts=pts, poses=poses)


def _gpu_graph(capi, ctx, graph, morton_on_odd=False):
    gs = []
    for i, (sm, (xyz, dist, w)) in enumerate(zip(graph["sms"], graph["pts"])):
        g = H.gpu_submap(capi, ctx, sm, i)
        g.set_points(capi.POINTS_VOXELS, xyz, dist, w,
                     capi.POINTS_SORT_MORTON if (morton_on_odd and i % 2) else capi.POINTS_KEEP_ORDER)
        gs.append(g)
    cfg = capi.default_config(registration_point_type=capi.POINTS_VOXELS, sampling_ratio=RATIO)
    cfs = [capi.RegistrationCostFunction(ctx, gs[a], gs[b], cfg) for a, b in PAIRS]
    return gs, cfs


def _eval_points(batch, poses, ctx):
    import torch
    R = batch.num_residuals()
    r = torch.full((R,), float("nan"), dtype=torch.float32, device="cuda:0")
    jo = torch.full((R, 4), float("nan"), dtype=torch.float32, device="cuda:0")
    je = torch.full((R, 4), float("nan"), dtype=torch.float32, device="cuda:0")
    torch.cuda.synchronize()
    status = batch.evaluate_points(poses, r.data_ptr(), jo.data_ptr(), je.data_ptr())
    ctx.synchronize()
    assert np.all(status == 0)
    return r.cpu().numpy(), jo.cpu().numpy(), je.cpu().numpy()


@pytest.mark.parametrize("morton", [False, True])
def test_batched_sampling_follows_the_shared_sampler_streams(capi, ctx, graph, morton):
    """Materialised rows of three successive batch evaluations + one drop-in Evaluate in between
    against the oracle driven by one mt19937 per reference point set, advanced in constraint order."""
    gs, cfs = _gpu_graph(capi, ctx, graph, morton_on_odd=morton)
    batch = capi.RegistrationBatch(ctx, cfs, PAIRS)
    ro = batch.row_offsets()
    layers = [H.oracle_layer(sm) for sm in graph["sms"]]
    engines = {a: orc.Mt19937(5489) for a in {p[0] for p in PAIRS}}
    cums = {a: sequential_cumsum(graph["pts"][a][2]) for a in engines}
    poses = graph["poses"]

    def oracle_rows(c):
        a, b = PAIRS[c]
        xyz, dist, w = graph["pts"][a]
        n = cfs[c].num_residuals()
        assert n == int(F(RATIO) * F(len(w)))
        idx = np.array([engines[a].weighted_draw(cums[a]) for _ in range(n)], np.int64)
        ok, r0, jo0, je0 = orc.reg_evaluate(layers[b], xyz, dist, w, poses[a], poses[b], sample_idx=idx)
        assert ok
        return r0, jo0, je0

    for call in range(2):
        r, jo, je = _eval_points(batch, poses, ctx)
        for c in range(len(PAIRS)):
            r0, jo0, je0 = oracle_rows(c)
            s = slice(ro[c], ro[c + 1])
            # f32 outputs of the same f64 values: exact
            assert np.array_equal(r[s], r0.astype(F)), (call, c)
            assert np.array_equal(jo[s], jo0.astype(F)) and np.array_equal(je[s], je0.astype(F)), (call, c)
    # the same pass in Ceres' own types (vgx_reg_batch_evaluate_points_f64) is one more evaluation of the batch: it draws
    # on, and every f64 is the oracle's
    import torch
    R = batch.num_residuals()
    r64 = torch.full((R,), float("nan"), dtype=torch.float64, device="cuda:0")
    jo64 = torch.full((R, 4), float("nan"), dtype=torch.float64, device="cuda:0")
    je64 = torch.full((R, 4), float("nan"), dtype=torch.float64, device="cuda:0")
    torch.cuda.synchronize()
    assert np.all(batch.evaluate_points_f64(poses, r64.data_ptr(), jo64.data_ptr(), je64.data_ptr()) == 0)
    ctx.synchronize()
    r64, jo64, je64 = r64.cpu().numpy(), jo64.cpu().numpy(), je64.cpu().numpy()
    for c in range(len(PAIRS)):
        r0, jo0, je0 = oracle_rows(c)
        s = slice(ro[c], ro[c + 1])
        assert np.array_equal(r64[s], r0) and np.array_equal(jo64[s], jo0) and np.array_equal(je64[s], je0), c
    # a drop-in Evaluate in between continues the SAME stream on the host ...
    c = 1
    n = cfs[c].num_residuals()
    r1, j1, j2 = np.zeros(n), np.zeros((n, 4)), np.zeros((n, 4))
    a, b = PAIRS[c]
    assert cfs[c].Evaluate([poses[a], poses[b]], r1, [j1, j2])
    r0, jo0, je0 = oracle_rows(c)
    assert np.array_equal(r1, r0) and np.array_equal(j1, jo0) and np.array_equal(j2, je0)
    # ... and the fused pass picks it up again on the device
    status, normal = batch.evaluate_normal(poses)
    assert np.all(status == 0)
    for c in range(len(PAIRS)):
        r0, jo0, je0 = oracle_rows(c)
        J = np.concatenate([jo0, je0], axis=1)
        cost, jtr, jtj = float(r0 @ r0), J.T @ r0, (J.T @ J)[np.triu_indices(8)]
        assert abs(normal[c, 0] - cost) <= 1e-6 * cost
        assert np.all(np.abs(normal[c, 1:9] - jtr) <= 1e-6 * np.abs(jtr).max())
        assert np.all(np.abs(normal[c, 9:] - jtj) <= 1e-6 * np.abs(jtj).max())
    # bitwise reproducible given the same stream position: rewind by rebuilding everything
    batch.destroy()
    for o in cfs + gs:
        o.destroy()


def test_two_batches_and_drop_in_calls_share_the_engines(capi, ctx, graph):
    """Two batches on the same point-set engines, evaluated alternately, with drop-in Evaluates in between
    and one batch destroyed on the way: every evaluation sees the engines' streams exactly as successive
    Evaluate calls in the same order would (RCF:113-122, weighted_sampler.h:36-39) -- the engine state
    travels between the host and the device as often as the callers alternate."""
    gs, cfs = _gpu_graph(capi, ctx, graph)
    sel_a, sel_b = [0, 1, 2, 3, 4], [2, 0, 4]                     # B: a subset, another order
    cfs_b = [capi.RegistrationCostFunction(ctx, gs[PAIRS[c][0]], gs[PAIRS[c][1]],
                                           capi.default_config(registration_point_type=capi.POINTS_VOXELS,
                                                               sampling_ratio=RATIO)) for c in sel_b]
    batch_a = capi.RegistrationBatch(ctx, cfs, PAIRS)
    batch_b = capi.RegistrationBatch(ctx, cfs_b, [PAIRS[c] for c in sel_b])
    layers = [H.oracle_layer(sm) for sm in graph["sms"]]
    engines = {a: orc.Mt19937(5489) for a in {p[0] for p in PAIRS}}
    cums = {a: sequential_cumsum(graph["pts"][a][2]) for a in engines}
    poses = graph["poses"]

    def oracle_rows(c):
        a, b = PAIRS[c]
        xyz, dist, w = graph["pts"][a]
        n = int(F(RATIO) * F(len(w)))
        idx = np.array([engines[a].weighted_draw(cums[a]) for _ in range(n)], np.int64)
        ok, r0, jo0, je0 = orc.reg_evaluate(layers[b], xyz, dist, w, poses[a], poses[b], sample_idx=idx)
        assert ok
        return r0, jo0, je0

    def check_batch(batch, sel, what):
        ro = batch.row_offsets()
        r, jo, je = _eval_points(batch, poses, ctx)
        for k, c in enumerate(sel):
            r0, jo0, je0 = oracle_rows(c)
            s = slice(ro[k], ro[k + 1])
            assert np.array_equal(r[s], r0.astype(F)), (what, c)
            assert np.array_equal(jo[s], jo0.astype(F)) and np.array_equal(je[s], je0.astype(F)), (what, c)

    def check_dropin(c, what):
        n = cfs[c].num_residuals()
        r1, j1, j2 = np.zeros(n), np.zeros((n, 4)), np.zeros((n, 4))
        a, b = PAIRS[c]
        assert cfs[c].Evaluate([poses[a], poses[b]], r1, [j1, j2])
        r0, jo0, je0 = oracle_rows(c)
        assert np.array_equal(r1, r0) and np.array_equal(j1, jo0) and np.array_equal(j2, je0), what

    check_batch(batch_a, sel_a, "A1")
    check_batch(batch_a, sel_a, "A2")
    check_batch(batch_b, sel_b, "B1")
    check_batch(batch_a, sel_a, "A3")
    check_dropin(3, "drop-in after A3")
    check_batch(batch_b, sel_b, "B2")
    check_batch(batch_b, sel_b, "B3")
    check_dropin(1, "drop-in after B3")
    check_dropin(2, "second drop-in in a row (engine already on the host)")
    batch_b.destroy()
    check_batch(batch_a, sel_a, "A4")
    check_batch(batch_a, sel_a, "A5")
    batch_a.destroy()
    for o in cfs + cfs_b + gs:
        o.destroy()


@pytest.mark.skipif(not ref_reg.available(), reason="oracle/_ref/libref_reg.so not built (needs /root/reference)")
def test_batched_sampling_equals_the_reference_source_called_in_the_same_order(capi, ctx, graph):
    """The same five sampling constraints through the reference's OWN RegistrationCostFunction
    (oracle/_ref: registration_cost_function.cpp compiled from /root/reference), its cost functions
    called in list order, twice: every residual and Jacobian entry of the batched HIP pass is the
    f32 rounding of the reference's f64 value."""
    gs, cfs = _gpu_graph(capi, ctx, graph)
    batch = capi.RegistrationBatch(ctx, cfs, PAIRS)
    ro = batch.row_offsets()
    poses = graph["poses"]
    refs = []
    for i, (sm, (xyz, dist, w)) in enumerate(zip(graph["sms"], graph["pts"])):
        R = ref_reg.Submap(i, poses[i], sm.voxel_size, sm.vps, sm.block_index, sm.tsdf_distance,
                           sm.tsdf_weight, sm.esdf_distance, sm.esdf_observed)
        R.set_points(ref_reg.POINTS_VOXELS, xyz, dist, w)        # same points, same order, same weights
        refs.append(R)
    ref_cfs = [ref_reg.RegistrationCostFunction(refs[a], refs[b], sampling_ratio=RATIO) for a, b in PAIRS]
    for call in range(2):
        r, jo, je = _eval_points(batch, poses, ctx)
        for c, (a, b) in enumerate(PAIRS):
            assert ref_cfs[c].num_residuals() == cfs[c].num_residuals()
            ok, r0, jo0, je0 = ref_cfs[c].Evaluate(poses[a], poses[b])
            assert ok
            s = slice(ro[c], ro[c + 1])
            assert np.array_equal(r[s], r0.astype(F)), (call, c)
            assert np.array_equal(jo[s], jo0.astype(F)) and np.array_equal(je[s], je0.astype(F)), (call, c)
    batch.destroy()
    for o in cfs + gs:
        o.destroy()


def test_device_mt19937_matches_the_host_engine_over_many_twists(capi, ctx, graph):
    """A long stream (many 624-word twists, ragged request sizes) generated on the device and read
    back through a drop-in Evaluate equals the host engine's: 10000th output of a default-seeded
    std::mt19937 is 4123659995 [C++ standard, rand.predef]."""
    eng = orc.Mt19937(5489)
    outs = [eng.next() for _ in range(10000)]
    assert outs[-1] == 4123659995
    # private engines (sampler_seed != 0) keep a constraint's stream independent of the point set's
    gs, _ = _gpu_graph(capi, ctx, graph)
    cfg = capi.default_config(registration_point_type=capi.POINTS_VOXELS, sampling_ratio=1.7, sampler_seed=77)
    cf = capi.RegistrationCostFunction(ctx, gs[0], gs[1], cfg)
    batch = capi.RegistrationBatch(ctx, [cf], [(0, 1)])
    xyz, dist, w = graph["pts"][0]
    layer = H.oracle_layer(graph["sms"][1])
    cum = sequential_cumsum(w)
    eng = orc.Mt19937(77)
    poses = graph["poses"]
    n = cf.num_residuals()
    assert n == int(F(1.7) * F(len(w)))
    for call in range(3):
        r, jo, je = _eval_points(batch, poses, ctx)
        idx = np.array([eng.weighted_draw(cum) for _ in range(n)], np.int64)
        ok, r0, jo0, je0 = orc.reg_evaluate(layer, xyz, dist, w, poses[0], poses[1], sample_idx=idx)
        assert np.array_equal(r, r0.astype(F)) and np.array_equal(jo, jo0.astype(F)), call
    batch.destroy()
    cf.destroy()
    for g in gs:
        g.destroy()


def test_replacing_points_invalidates_cost_functions_and_batches(capi, ctx, graph):
    """a cost function / batch holds raw device pointers into the reference submap's
    point arrays; re-uploading or re-extracting the points must turn later evaluations into a
    clean VGX_ERR_INVALID instead of a read of freed memory."""
    gs, cfs = _gpu_graph(capi, ctx, graph)
    batch = capi.RegistrationBatch(ctx, cfs, PAIRS)
    xyz, dist, w = graph["pts"][0]
    gs[0].set_points(capi.POINTS_VOXELS, xyz[:100], dist[:100], w[:100])
    with pytest.raises(capi.VgxError):
        batch.evaluate_normal(graph["poses"])
    n = cfs[0].num_residuals()
    with pytest.raises(capi.VgxError):
        cfs[0].Evaluate([graph["poses"][0], graph["poses"][1]], np.zeros(n), None)
    with pytest.raises(capi.VgxError):
        capi.RegistrationBatch(ctx, cfs, PAIRS)
    # constraints whose reference points were left alone still work
    ok = cfs[2].Evaluate([graph["poses"][1], graph["poses"][0]], np.zeros(cfs[2].num_residuals()), None)
    assert ok
    batch.destroy()
    for o in cfs + gs:
        o.destroy()
