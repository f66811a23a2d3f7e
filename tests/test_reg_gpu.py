"""GPU parity tests of the REG path: HIP kernels (through the C ABI) vs the CPU
oracle on identical seeded inputs.  Tolerance: north_star's 1e-4 relative on
residuals and Jacobians (tests/helpers.py defines "relative")."""
import numpy as np
import pytest

from oracle import pyoracle as orc
from oracle import synth
from tests import helpers as H

pytestmark = pytest.mark.gpu
F = np.float32


@pytest.fixture(scope="module")
def capi():
    from voxgraph_amd import capi
    return capi


@pytest.fixture(scope="module")
def ctx(capi):
    c = capi.Context(0)
    yield c
    c.close()


@pytest.fixture(scope="module")
def cfg1(capi, ctx):
    """BASELINE config 1: duplicated 64^3 sphere+ground submap."""
    sm, _ = synth.config1_pair()
    g = H.gpu_submap(capi, ctx, sm)
    xyz, dist, w = H.oracle_points(sm)
    return sm, g, H.oracle_layer(sm), (xyz, dist, w)


def _gpu_eval(cf, ref_pose, read_pose, want_ref=True, want_read=True, want_jac=True):
    n = cf.num_residuals()
    r = np.full(n, np.nan)
    jo = np.full((n, 4), np.nan) if (want_jac and want_ref) else None
    je = np.full((n, 4), np.nan) if (want_jac and want_read) else None
    ok = cf.Evaluate([ref_pose, read_pose], r, [jo, je] if want_jac else None)
    return ok, r, jo, je


def test_device_extraction_is_bit_exact(capi, ctx, cfg1):
    """voxgraph_submap.cpp:144-201 on the device == oracle, same order, same bits."""
    sm, g, _, (xyz, dist, w) = cfg1
    n = g.extract_voxel_points(1.0, 0.3, True)
    assert n == len(w)
    gx, gd, gw = g.download_points(capi.POINTS_VOXELS)
    assert np.array_equal(gx, xyz) and np.array_equal(gd, dist) and np.array_equal(gw, w)
    assert np.array_equal(g.point_order(capi.POINTS_VOXELS), np.arange(n))
    # TSDF-distance variant and a different filter
    x2, d2, w2 = H.oracle_points(sm, use_esdf=False, min_w=0.5, max_d=0.17)
    g2 = H.gpu_submap(capi, ctx, sm, 7)
    assert g2.extract_voxel_points(0.5, 0.17, False) == len(w2)
    a, b, c = g2.download_points(capi.POINTS_VOXELS)
    assert np.array_equal(a, x2) and np.array_equal(b, d2) and np.array_equal(c, w2)
    g2.destroy()


def test_evaluate_matches_oracle_on_reference_test_grid(capi, ctx, cfg1):
    """Drop-in Evaluate vs oracle over the reference's perturbation grid
    (registration_test_bench.yaml:9-13)."""
    sm, g, layer, (xyz, dist, w) = cfg1
    g.extract_voxel_points()
    cf = capi.RegistrationCostFunction(
        ctx, g, g, capi.default_config(registration_point_type=capi.POINTS_VOXELS))
    assert cf.num_residuals() == len(w)
    worst = 0.0
    grid = H.test_bench_grid(sm.voxel_size)
    for k, pert in enumerate(grid[::3] + [np.zeros(4)]):
        ref_pose = np.array([0.02, -0.01, 0.03, 0.01])
        read_pose = ref_pose + pert
        ok, r, jo, je = _gpu_eval(cf, ref_pose, read_pose)
        ok0, r0, jo0, je0 = orc.reg_evaluate(layer, xyz, dist, w, ref_pose, read_pose)
        assert ok and ok0
        worst = max(worst, H.assert_parity(r, r0, f"residual[{k}]"),
                    H.assert_parity(jo, jo0, f"jac_ref[{k}]"),
                    H.assert_parity(je, je0, f"jac_read[{k}]"))
    print("worst relative error over the grid:", worst)
    cf.destroy()


def test_identical_poses_give_exactly_zero_residuals(capi, ctx, cfg1):
    """Duplicate submap at the same pose: every point is a voxel centre of the
    reading grid, Delta == 0, so r == (d - d) w == 0 exactly (size-independent)."""
    sm, g, layer, (xyz, dist, w) = cfg1
    g.extract_voxel_points()
    cf = capi.RegistrationCostFunction(
        ctx, g, g, capi.default_config(registration_point_type=capi.POINTS_VOXELS))
    ok, r, jo, je = _gpu_eval(cf, np.zeros(4), np.zeros(4))
    assert ok and np.all(r == 0.0)
    cf.destroy()


def test_morton_order_is_a_permutation_of_the_same_rows(capi, ctx, cfg1):
    sm, _, layer, (xyz, dist, w) = cfg1
    g = H.gpu_submap(capi, ctx, sm, 3)
    g.set_points(capi.POINTS_VOXELS, xyz, dist, w, capi.POINTS_SORT_MORTON)
    order = g.point_order(capi.POINTS_VOXELS)
    assert np.array_equal(np.sort(order), np.arange(len(w)))
    cf = capi.RegistrationCostFunction(
        ctx, g, g, capi.default_config(registration_point_type=capi.POINTS_VOXELS))
    ref_pose = np.array([0.05, 0.02, -0.03, 0.04])
    read_pose = np.array([-0.02, 0.0, 0.01, -0.03])
    ok, r, jo, je = _gpu_eval(cf, ref_pose, read_pose)
    ok0, r0, jo0, je0 = orc.reg_evaluate(layer, xyz, dist, w, ref_pose, read_pose)
    assert ok and ok0
    H.assert_parity(r, r0[order], "residual")
    H.assert_parity(jo, jo0[order], "jac_ref")
    H.assert_parity(je, je0[order], "jac_read")
    cf.destroy()
    g.destroy()


def test_null_jacobian_blocks_and_no_jacobians(capi, ctx, cfg1):
    """jacobians == nullptr (.cpp:179) and jacobians[k] == nullptr (.cpp:254,261)."""
    sm, g, layer, (xyz, dist, w) = cfg1
    g.extract_voxel_points()
    cf = capi.RegistrationCostFunction(
        ctx, g, g, capi.default_config(registration_point_type=capi.POINTS_VOXELS))
    a, b = np.array([0.03, 0.0, 0.0, 0.02]), np.array([0.0, 0.02, 0.01, 0.0])
    ok, r_full, jo_full, je_full = _gpu_eval(cf, a, b)
    ok, r, jo, je = _gpu_eval(cf, a, b, want_jac=False)
    assert ok and np.array_equal(r, r_full)
    ok, r, jo, je = _gpu_eval(cf, a, b, want_ref=False)
    assert ok and jo is None and np.array_equal(je, je_full) and np.array_equal(r, r_full)
    ok, r, jo, je = _gpu_eval(cf, a, b, want_read=False)
    assert ok and je is None and np.array_equal(jo, jo_full)
    cf.destroy()


def test_sparse_submap_missing_blocks_and_no_correspondence_cost(capi, ctx):
    """Missing blocks / unobserved neighbours => w * no_correspondence_cost and zero
    Jacobian rows (.cpp:164-167,240-243); negative block indices; partial overlap."""
    sdf = synth.sphere_ground_sdf((0.3, -0.2, 0.1), 1.5, -1.0)
    ref = synth.make_submap(sdf, 0.1, 16, (-2, -2, -2), (4, 4, 4), trunc=0.3, esdf_max=0.8,
                            drop_empty_blocks=True)
    read = synth.make_submap(sdf, 0.1, 16, (-1, -2, -1), (3, 3, 3), trunc=0.3, esdf_max=0.8,
                             pose=(0.4, 0.1, -0.2, 0.3), drop_empty_blocks=True)
    assert 0 < ref.n_blocks < 64
    g_ref, g_read = H.gpu_submap(capi, ctx, ref, 0), H.gpu_submap(capi, ctx, read, 1)
    xyz, dist, w = H.oracle_points(ref)
    w = (w * np.random.default_rng(5).uniform(0.2, 1.0, len(w))).astype(F)   # varied weights
    g_ref.set_points(capi.POINTS_VOXELS, xyz, dist, w)
    layer = H.oracle_layer(read)
    cfg = capi.default_config(registration_point_type=capi.POINTS_VOXELS,
                              no_correspondence_cost=0.37)
    cf = capi.RegistrationCostFunction(ctx, g_ref, g_read, cfg)
    ref_pose = np.array([0.0, 0.0, 0.0, 0.0])
    read_pose = np.array([0.45, 0.05, -0.22, 0.27])
    ok, r, jo, je = _gpu_eval(cf, ref_pose, read_pose)
    ok0, r0, jo0, je0 = orc.reg_evaluate(layer, xyz, dist, w, ref_pose, read_pose,
                                         no_correspondence_cost=0.37)
    assert ok and ok0
    no_corr = np.all(jo0 == 0, axis=1) & np.all(je0 == 0, axis=1)
    assert 0.02 < no_corr.mean() < 0.98, no_corr.mean()
    assert np.array_equal(np.all(jo == 0, axis=1) & np.all(je == 0, axis=1), no_corr)
    H.assert_parity(r, r0, "residual")
    H.assert_parity(jo, jo0, "jac_ref")
    H.assert_parity(je, je0, "jac_read")
    for o in (cf, g_ref, g_read):
        o.destroy()


def test_tsdf_distance_mode_and_vps8(capi, ctx):
    """use_esdf_distance == false samples the TSDF layer, valid iff weight > 0
    (.cpp:143-153); voxels_per_side 8."""
    sdf = synth.sphere_ground_sdf((1.0, 1.0, 1.0), 0.8, 0.3)
    sm = synth.make_submap(sdf, 0.05, 8, (0, 0, 0), (5, 5, 5), trunc=0.15, esdf_max=0.5)
    g = H.gpu_submap(capi, ctx, sm)
    xyz, dist, w = H.oracle_points(sm, use_esdf=False, max_d=0.1)
    assert g.extract_voxel_points(1.0, 0.1, False) == len(w)
    layer = H.oracle_layer(sm, use_esdf=False)
    cfg = capi.default_config(registration_point_type=capi.POINTS_VOXELS, use_esdf_distance=0)
    cf = capi.RegistrationCostFunction(ctx, g, g, cfg)
    a, b = np.array([0.01, 0.0, 0.02, 0.05]), np.array([0.0, 0.03, 0.0, -0.02])
    ok, r, jo, je = _gpu_eval(cf, a, b)
    ok0, r0, jo0, je0 = orc.reg_evaluate(layer, xyz, dist, w, a, b)
    assert ok and ok0 and np.any(np.all(jo0 == 0, axis=1)) and np.any(jo0 != 0)
    H.assert_parity(r, r0, "residual")
    H.assert_parity(jo, jo0, "jac_ref")
    H.assert_parity(je, je0, "jac_read")
    cf.destroy()
    g.destroy()


def test_zero_weight_sum_returns_false_and_empty_point_set(capi, ctx, cfg1):
    """.cpp:273"""
    sm, _, _, (xyz, dist, w) = cfg1
    g = H.gpu_submap(capi, ctx, sm, 9)
    g.set_points(capi.POINTS_VOXELS, xyz[:100], dist[:100], np.zeros(100, F))
    cf = capi.RegistrationCostFunction(
        ctx, g, g, capi.default_config(registration_point_type=capi.POINTS_VOXELS))
    ok, *_ = _gpu_eval(cf, np.zeros(4), np.zeros(4))
    assert ok is False
    cf.destroy()
    g.set_points(capi.POINTS_ISOSURFACE, xyz[:0], dist[:0], w[:0])
    cf = capi.RegistrationCostFunction(ctx, g, g, capi.default_config())
    assert cf.num_residuals() == 0
    assert _gpu_eval(cf, np.zeros(4), np.zeros(4))[0] is False     # sum of no weights == 0
    cf.destroy()
    # unfinished submap (no points of the requested type) is an error, not a crash
    g3 = H.gpu_submap(capi, ctx, sm, 10)
    with pytest.raises(capi.VgxError):
        capi.RegistrationCostFunction(ctx, g3, g3, capi.default_config())
    g3.destroy()
    g.destroy()


def test_sampling_mode_reproduces_the_weighted_sampler_stream(capi, ctx, cfg1):
    """sampling_ratio != -1 (.cpp:46-50,115-122): default-seeded mt19937, draws
    proportional to weight, weight := 1; two consecutive Evaluates continue the
    stream like the reference's mutable sampler."""
    sm, _, layer, (xyz, dist, w) = cfg1
    rng = np.random.default_rng(11)
    w2 = (w * rng.uniform(0.1, 1.0, len(w))).astype(F)
    for flags in (capi.POINTS_KEEP_ORDER, capi.POINTS_SORT_MORTON):
        g = H.gpu_submap(capi, ctx, sm, 20)
        g.set_points(capi.POINTS_VOXELS, xyz, dist, w2, flags)
        cfg = capi.default_config(registration_point_type=capi.POINTS_VOXELS, sampling_ratio=0.05)
        cf = capi.RegistrationCostFunction(ctx, g, g, cfg)
        n = cf.num_residuals()
        assert n == int(np.float32(0.05) * np.float32(len(w2)))
        cum = np.cumsum(w2.astype(np.float64))
        # addItem accumulates sequentially in double (weighted_sampler_inl.h:5-16)
        acc, cum_seq = 0.0, np.zeros(len(w2))
        for i, v in enumerate(w2.astype(np.float64)):
            acc = v if i == 0 else acc + v
            cum_seq[i] = acc
        eng = orc.Mt19937(5489)
        a, b = np.array([0.02, 0.01, 0.0, 0.03]), np.array([0.0, 0.0, 0.02, -0.01])
        for call in range(2):
            idx = np.array([eng.weighted_draw(cum_seq) for _ in range(n)], np.int64)
            ok, r, jo, je = _gpu_eval(cf, a, b)
            ok0, r0, jo0, je0 = orc.reg_evaluate(layer, xyz, dist, w2, a, b, sample_idx=idx)
            assert ok and ok0
            H.assert_parity(r, r0, f"residual call {call}")
            H.assert_parity(jo, jo0, f"jac_ref call {call}")
            H.assert_parity(je, je0, f"jac_read call {call}")
        cf.destroy()
        g.destroy()


def _torch_buf(n, dtype):
    import torch
    return torch.full((n,), float("nan"), dtype=dtype, device="cuda:0")


def test_device_f32_outputs(capi, ctx, cfg1):
    """The 88 B/evaluation form: f32 results left on the device."""
    import torch
    sm, g, layer, (xyz, dist, w) = cfg1
    g.extract_voxel_points()
    cf = capi.RegistrationCostFunction(
        ctx, g, g, capi.default_config(registration_point_type=capi.POINTS_VOXELS))
    n = cf.num_residuals()
    r, jo, je = _torch_buf(n, torch.float32), _torch_buf(4 * n, torch.float32), _torch_buf(4 * n, torch.float32)
    a, b = np.array([0.03, -0.02, 0.01, 0.02]), np.array([0.0, 0.01, 0.0, -0.02])
    torch.cuda.synchronize()
    assert cf.evaluate_device_f32(a, b, r.data_ptr(), jo.data_ptr(), je.data_ptr())
    ctx.synchronize()
    ok0, r0, jo0, je0 = orc.reg_evaluate(layer, xyz, dist, w, a, b)
    H.assert_parity(r.cpu().numpy(), r0, "residual")
    H.assert_parity(jo.cpu().numpy().reshape(n, 4), jo0, "jac_ref")
    H.assert_parity(je.cpu().numpy().reshape(n, 4), je0, "jac_read")
    cf.destroy()


@pytest.fixture(scope="module")
def small_graph(capi, ctx):
    """4 overlapping submaps of one scene, 5 constraints (both directions of one pair)."""
    sdf = synth.sphere_ground_sdf((1.6, 1.6, 1.2), 1.0, 0.35)
    poses_true = [(0, 0, 0, 0), (0.8, 0.1, 0.0, 0.1), (0.1, 0.9, 0.05, -0.15), (0.9, 0.8, 0.0, 0.2)]
    sms, gs, layers, pts = [], [], [], []
    for i, p in enumerate(poses_true):
        sm = synth.make_submap(sdf, 0.1, 16, (0, 0, 0), (2, 2, 2), trunc=0.3, esdf_max=1.0,
                               pose=p, drop_empty_blocks=True)
        g = H.gpu_submap(capi, ctx, sm, i)
        xyz, dist, w = H.oracle_points(sm)
        g.set_points(capi.POINTS_VOXELS, xyz, dist, w,
                     capi.POINTS_SORT_MORTON if i % 2 else capi.POINTS_KEEP_ORDER)
        order = g.point_order(capi.POINTS_VOXELS)
        sms.append(sm), gs.append(g), layers.append(H.oracle_layer(sm))
        pts.append((xyz[order], dist[order], w[order]))
    pairs = [(0, 1), (1, 0), (0, 2), (1, 3), (2, 3)]
    cfg = capi.default_config(registration_point_type=capi.POINTS_VOXELS)
    cfs = [capi.RegistrationCostFunction(ctx, gs[a], gs[b], cfg) for a, b in pairs]
    rng = np.random.default_rng(2)
    poses = np.array(poses_true, np.float64) + rng.normal(0, 0.03, (4, 4))
    yield dict(gs=gs, layers=layers, pts=pts, pairs=pairs, cfs=cfs, poses=poses)
    for o in cfs + gs:
        o.destroy()


def test_batch_points_match_per_constraint_oracle(capi, ctx, small_graph):
    import torch
    G = small_graph
    batch = capi.RegistrationBatch(ctx, G["cfs"], G["pairs"])
    ro = batch.row_offsets()
    R = batch.num_residuals()
    assert ro[-1] == R == sum(cf.num_residuals() for cf in G["cfs"])
    r, jo, je = _torch_buf(R, torch.float32), _torch_buf(4 * R, torch.float32), _torch_buf(4 * R, torch.float32)
    torch.cuda.synchronize()
    status = batch.evaluate_points(G["poses"], r.data_ptr(), jo.data_ptr(), je.data_ptr())
    ctx.synchronize()
    assert np.all(status == 0)
    r, jo, je = r.cpu().numpy(), jo.cpu().numpy().reshape(R, 4), je.cpu().numpy().reshape(R, 4)
    for c, (a, b) in enumerate(G["pairs"]):
        xyz, dist, w = G["pts"][a]
        ok0, r0, jo0, je0 = orc.reg_evaluate(G["layers"][b], xyz, dist, w, G["poses"][a], G["poses"][b])
        s = slice(ro[c], ro[c + 1])
        H.assert_parity(r[s], r0, f"residual c{c}")
        H.assert_parity(jo[s], jo0, f"jac_ref c{c}")
        H.assert_parity(je[s], je0, f"jac_read c{c}")
    batch.destroy()


def test_choose_outputs_times_candidates_and_leaves_valid_rows(capi, ctx, small_graph):
    """vgx_reg_batch_choose_outputs (placement by measurement): every trial is a real launch of the batch, the indices
    it returns are candidates, and whichever arrays it picks hold the rows a plain evaluate_points writes"""
    import torch
    G = small_graph
    batch = capi.RegistrationBatch(ctx, G["cfs"], G["pairs"])
    R = batch.num_residuals()
    n = 3
    cand = [(_torch_buf(R, torch.float32), _torch_buf(4 * R, torch.float32), _torch_buf(4 * R, torch.float32)) for _ in range(n)]
    ref = (_torch_buf(R, torch.float32), _torch_buf(4 * R, torch.float32), _torch_buf(4 * R, torch.float32))
    torch.cuda.synchronize()
    chosen, ms, trials = batch.choose_outputs(G["poses"], [c[0].data_ptr() for c in cand], [c[1].data_ptr() for c in cand],
                                              [c[2].data_ptr() for c in cand], launches=2)
    assert len(chosen) == 3 and all(0 <= k < n for k in chosen)
    assert ms > 0 and len(trials) == 4 * n
    assert all(t > 0 for t in trials[:n])                                   # the sets
    for which in range(3):                                                  # then jac_read, jac_ref, residuals
        block = trials[n + which * n: n + (which + 1) * n]
        assert sum(1 for t in block if t == -1.0) == 1 and all(t > 0 or t == -1.0 for t in block)
    assert ms <= min(t for t in trials if t > 0) * 1.006     # (a change is adopted only when it gains 0.5 %)
    batch.evaluate_points(G["poses"], ref[0].data_ptr(), ref[1].data_ptr(), ref[2].data_ptr())
    ctx.synchronize()
    for which, k in enumerate(chosen):
        assert torch.equal(cand[k][which], ref[which])
    # without Jacobians: only the residual arrays are candidates
    chosen2, ms2, trials2 = batch.choose_outputs(G["poses"], [c[0].data_ptr() for c in cand], None, None, launches=1)
    assert 0 <= chosen2[0] < n and ms2 > 0
    with pytest.raises(Exception):
        batch.choose_outputs(G["poses"], [cand[0][0].data_ptr(), 0], None, None)
    batch.destroy()
    # a SAMPLING batch is refused: its trial evaluations would advance the reference point sets' engines
    scfg = capi.default_config(registration_point_type=capi.POINTS_VOXELS, sampling_ratio=0.5)
    scfs = [capi.RegistrationCostFunction(ctx, G["gs"][a], G["gs"][b], scfg) for a, b in G["pairs"]]
    sbatch = capi.RegistrationBatch(ctx, scfs, G["pairs"])
    with pytest.raises(Exception, match="sampling"):
        sbatch.choose_outputs(G["poses"], [c[0].data_ptr() for c in cand], [c[1].data_ptr() for c in cand],
                              [c[2].data_ptr() for c in cand], launches=1)
    sbatch.destroy()
    for cf in scfs:
        cf.destroy()


def test_alloc_outputs_returns_chosen_arrays_of_the_library(capi, ctx, small_graph):
    """vgx_reg_batch_alloc_outputs: the library allocates the candidate sets, chooses among them by timing the batch's own launch
    (vgx_reg_batch_choose_outputs), frees the rest and hands back three arrays: they hold the rows any other arrays would; a
    missing Jacobian block is left out; released with vgx_reg_batch_free_outputs"""
    import ctypes
    import torch
    G = small_graph
    batch = capi.RegistrationBatch(ctx, G["cfs"], G["pairs"])
    R = batch.num_residuals()
    r, jo, je = _torch_buf(R, torch.float32), _torch_buf(4 * R, torch.float32), _torch_buf(4 * R, torch.float32)
    torch.cuda.synchronize()
    batch.evaluate_points(G["poses"], r.data_ptr(), jo.data_ptr(), je.data_ptr())
    pr, pjo, pje, ms = batch.alloc_outputs(G["poses"], n_candidates=3)
    assert pr and pjo and pje and ms > 0 and pr % 16 == 0 and pjo % 16 == 0 and pje % 16 == 0
    batch.evaluate_points(G["poses"], pr, pjo, pje)
    ctx.synchronize()
    got = [torch.empty(n, dtype=torch.float32, device="cuda") for n in (R, 4 * R, 4 * R)]
    hip = ctypes.CDLL("libamdhip64.so")
    for dst, src, n in zip(got, (pr, pjo, pje), (R, 4 * R, 4 * R)):
        assert hip.hipMemcpy(ctypes.c_void_p(dst.data_ptr()), ctypes.c_void_p(src), ctypes.c_size_t(4 * n), 3) == 0   # device to device
    torch.cuda.synchronize()
    for a, b in zip(got, (r, jo, je)):
        assert torch.equal(a.view(torch.int32), b.view(torch.int32))
    batch.free_outputs(pr, pjo, pje)
    pr2, pjo2, pje2, _ = batch.alloc_outputs(G["poses"], n_candidates=2, want_jac_ref=False)     # a constant block: no array for it
    assert pr2 and not pjo2 and pje2
    batch.evaluate_points(G["poses"], pr2, None, pje2)
    ctx.synchronize()
    batch.free_outputs(pr2, 0, pje2)
    batch.destroy()


def test_blocked_output_layout_holds_the_same_rows(capi, ctx, small_graph):
    """vgx_reg_batch_evaluate_points_blocked: ONE array of 36 KiB tile blocks ([r x 1024][jac_ref x 1024][jac_read x 1024],
    every constraint padded to whole blocks) instead of three arrays -- the same kernel, so the same values bit for bit;
    rows of a constraint's last block beyond its residuals are not written"""
    import torch
    G = small_graph
    batch = capi.RegistrationBatch(ctx, G["cfs"], G["pairs"])
    R = batch.num_residuals()
    ro = batch.row_offsets()
    nbytes, rows, first = batch.blocked_layout()
    assert rows == 1024 and nbytes == int(first[-1]) * rows * 36
    assert all(first[c + 1] - first[c] == -(-(ro[c + 1] - ro[c]) // rows) for c in range(len(G["pairs"])))
    r, jo, je = _torch_buf(R, torch.float32), _torch_buf(4 * R, torch.float32), _torch_buf(4 * R, torch.float32)
    blocks = torch.full((nbytes // 4,), float("nan"), dtype=torch.float32, device="cuda")
    torch.cuda.synchronize()
    st0 = batch.evaluate_points(G["poses"], r.data_ptr(), jo.data_ptr(), je.data_ptr())
    st1 = batch.evaluate_points_blocked(G["poses"], blocks.data_ptr())
    ctx.synchronize()
    assert np.array_equal(st0, st1)
    r, jo, je = r.cpu().numpy(), jo.cpu().numpy().reshape(R, 4), je.cpu().numpy().reshape(R, 4)
    B = blocks.cpu().numpy().reshape(-1, 9 * rows)
    written = 0
    for c in range(len(G["pairs"])):
        n = int(ro[c + 1] - ro[c])
        for k0 in range(0, n, rows):
            blk = B[int(first[c]) + k0 // rows]
            m = min(rows, n - k0)
            s = slice(int(ro[c]) + k0, int(ro[c]) + k0 + m)
            assert np.array_equal(blk[:m].view(np.uint32), r[s].view(np.uint32))
            assert np.array_equal(blk[rows:rows + 4 * rows].reshape(rows, 4)[:m].view(np.uint32), jo[s].view(np.uint32))
            assert np.array_equal(blk[5 * rows:].reshape(rows, 4)[:m].view(np.uint32), je[s].view(np.uint32))
            assert np.isnan(blk[m:rows]).all() and np.isnan(blk[rows:5 * rows].reshape(rows, 4)[m:]).all()   # padding untouched
            written += m
    assert written == R
    with pytest.raises(Exception):
        batch.evaluate_points_blocked(G["poses"], blocks.data_ptr() + 4)     # not 16-byte aligned
    batch.destroy()


def test_batched_f64_rows_are_the_reference_f64_rows(capi, ctx, small_graph):
    """vgx_reg_batch_evaluate_points_f64: the materialising pass in Ceres' own types (SURVEY.md 8d's 124-byte variant) -- every
    f64 EQUAL to the oracle's (which is pinned to the reference's own source: tests/test_ref_pin.py) and to the drop-in
    vgx_reg_evaluate's for the same constraint; its f32 rounding is the f32 pass's row, bit for bit; a null Jacobian block
    is honoured; a misaligned Jacobian array is refused"""
    import torch
    G = small_graph
    batch = capi.RegistrationBatch(ctx, G["cfs"], G["pairs"])
    ro, R = batch.row_offsets(), batch.num_residuals()
    r, jo, je = _torch_buf(R, torch.float64), _torch_buf(4 * R, torch.float64), _torch_buf(4 * R, torch.float64)
    r32, jo32, je32 = _torch_buf(R, torch.float32), _torch_buf(4 * R, torch.float32), _torch_buf(4 * R, torch.float32)
    torch.cuda.synchronize()
    st = batch.evaluate_points_f64(G["poses"], r.data_ptr(), jo.data_ptr(), je.data_ptr())
    st32 = batch.evaluate_points(G["poses"], r32.data_ptr(), jo32.data_ptr(), je32.data_ptr())
    ctx.synchronize()
    assert np.all(st == 0) and np.array_equal(st, st32)
    rr, jjo, jje = r.cpu().numpy(), jo.cpu().numpy().reshape(R, 4), je.cpu().numpy().reshape(R, 4)
    assert np.array_equal(rr.astype(F).view(np.uint32), r32.cpu().numpy().view(np.uint32))
    assert np.array_equal(jjo.astype(F).view(np.uint32), jo32.cpu().numpy().reshape(R, 4).view(np.uint32))
    assert np.array_equal(jje.astype(F).view(np.uint32), je32.cpu().numpy().reshape(R, 4).view(np.uint32))
    rows = 0
    for c, (a, b) in enumerate(G["pairs"]):
        xyz, dist, w = G["pts"][a]
        ok0, r0, jo0, je0 = orc.reg_evaluate(G["layers"][b], xyz, dist, w, G["poses"][a], G["poses"][b])
        s = slice(ro[c], ro[c + 1])
        assert ok0 and np.array_equal(rr[s], r0) and np.array_equal(jjo[s], jo0) and np.array_equal(jje[s], je0), c
        ok1, r1, jo1, je1 = _gpu_eval(G["cfs"][c], G["poses"][a], G["poses"][b])
        assert ok1 and np.array_equal(rr[s], r1) and np.array_equal(jjo[s], jo1) and np.array_equal(jje[s], je1), c
        rows += len(r0)
    assert rows == R and np.count_nonzero(jjo) > 0
    # jacobians[0] == nullptr (a constant block, pose_graph_interface.cpp:30-32): residuals and the other block as before
    je2 = _torch_buf(4 * R, torch.float64)
    r2 = _torch_buf(R, torch.float64)
    torch.cuda.synchronize()
    batch.evaluate_points_f64(G["poses"], r2.data_ptr(), None, je2.data_ptr())
    ctx.synchronize()
    assert np.array_equal(r2.cpu().numpy(), rr) and np.array_equal(je2.cpu().numpy().reshape(R, 4), jje)
    with pytest.raises(Exception, match="aligned"):
        batch.evaluate_points_f64(G["poses"], r.data_ptr(), jo.data_ptr() + 16, je.data_ptr())
    batch.destroy()


def test_rows_kept_by_the_batch_and_fetched_per_constraint(capi, ctx, small_graph):
    """vgx_reg_batch_evaluate_rows_f64 + vgx_reg_batch_fetch_rows_f64 (SURVEY.md 8b's vgx_reg_fetch): one launch for the whole
    list, every constraint's slice the drop-in vgx_reg_evaluate's values bit for bit; a Jacobian block that was not asked for is
    refused; so is a fetch without an evaluation"""
    G = small_graph
    batch = capi.RegistrationBatch(ctx, G["cfs"], G["pairs"])
    with pytest.raises(Exception, match="no rows evaluation"):
        batch.fetch_rows_f64(0, G["cfs"][0].num_residuals())
    for poses in (G["poses"], G["poses"] + 0.01):
        assert np.all(batch.evaluate_rows_f64(poses) == 0)
        for c, (a, b) in enumerate(G["pairs"]):
            n = G["cfs"][c].num_residuals()
            r, jo, je = batch.fetch_rows_f64(c, n)
            ok1, r1, jo1, je1 = _gpu_eval(G["cfs"][c], poses[a], poses[b])
            assert ok1 and np.array_equal(r, r1) and np.array_equal(jo, jo1) and np.array_equal(je, je1), c
    # jacobians == nullptr (a cost-only evaluation): residuals alone; a Jacobian fetch is then an error, not stale numbers
    assert np.all(batch.evaluate_rows_f64(G["poses"], want_jac_ref=False, want_jac_read=False) == 0)
    n0 = G["cfs"][0].num_residuals()
    r, _, _ = batch.fetch_rows_f64(0, n0, want_jac_ref=False, want_jac_read=False)
    a, b = G["pairs"][0]
    assert np.array_equal(r, _gpu_eval(G["cfs"][0], G["poses"][a], G["poses"][b])[1])
    with pytest.raises(Exception, match="no such Jacobian"):
        batch.fetch_rows_f64(0, n0)
    with pytest.raises(Exception):
        batch.fetch_rows_f64(len(G["pairs"]), 1)
    batch.destroy()


def test_cost_only_pass_is_the_full_pass_cost_bit_for_bit(capi, ctx, small_graph):
    """vgx_reg_batch_evaluate_cost (what Ceres asks for at every trial step: `jacobians == nullptr`,
    registration_cost_function.cpp:179): the same f32 operations in the same order through the same reduction tree, so the
    cost a step is accepted on is the very number the full evaluation at that point reports -- and within 1e-6 of the
    oracle's; with a no-correspondence cost, on the device, reproducibly, and interleaved with full passes"""
    import torch
    G = small_graph
    batch = capi.RegistrationBatch(ctx, G["cfs"], G["pairs"])
    rng = np.random.default_rng(2)
    for trial in range(4):
        poses = G["poses"] + (rng.normal(0, [0.1, 0.1, 0.05, 0.02], G["poses"].shape) if trial else 0)
        st_n, normal = batch.evaluate_normal(poses)
        st_c, cost = batch.evaluate_cost(poses)
        assert np.array_equal(st_n, st_c)
        assert np.array_equal(cost.view(np.uint64), normal[:, 0].copy().view(np.uint64)), (trial, cost, normal[:, 0])
        _, cost2 = batch.evaluate_cost(poses)
        assert np.array_equal(cost.view(np.uint64), cost2.view(np.uint64))
    for c, (a, b) in enumerate(G["pairs"]):
        xyz, dist, w = G["pts"][a]
        ok, want, _, _ = orc.reg_evaluate_normal(G["layers"][b], xyz, dist, w, poses[a], poses[b])
        assert abs(cost[c] - want) <= 1e-6 * want
    # into a caller's device array, nothing brought to the host
    d_cost = torch.full((batch.n,), float("nan"), dtype=torch.float64, device="cuda:0")
    torch.cuda.synchronize()
    batch.evaluate_cost(poses, d_cost=d_cost.data_ptr(), to_host=False)
    ctx.synchronize()
    assert np.array_equal(d_cost.cpu().numpy().view(np.uint64), cost.view(np.uint64))
    batch.destroy()
    # no_correspondence_cost != 0: misses count (RCF:165-166), nothing is culled
    cfs = [capi.RegistrationCostFunction(ctx, G["gs"][a], G["gs"][b],
                                         capi.default_config(registration_point_type=capi.POINTS_VOXELS, no_correspondence_cost=0.3))
           for a, b in G["pairs"]]
    b2 = capi.RegistrationBatch(ctx, cfs, G["pairs"])
    _, normal = b2.evaluate_normal(poses)
    _, cost = b2.evaluate_cost(poses)
    assert np.array_equal(cost.view(np.uint64), normal[:, 0].copy().view(np.uint64)) and np.all(cost > 0)
    b2.destroy()
    for cf in cfs:
        cf.destroy()


def test_batch_normal_equations_and_assembly(capi, ctx, small_graph):
    import torch
    G = small_graph
    n_nodes = 5     # one more node than the batch touches
    batch = capi.RegistrationBatch(ctx, G["cfs"], G["pairs"])
    status, normal = batch.evaluate_normal(G["poses"])
    assert np.all(status == 0)
    want = np.zeros_like(normal)
    for c, (a, b) in enumerate(G["pairs"]):
        xyz, dist, w = G["pts"][a]
        ok, cost, jtr, jtj = orc.reg_evaluate_normal(G["layers"][b], xyz, dist, w, G["poses"][a], G["poses"][b])
        want[c] = np.concatenate([[cost], jtr, jtj])
        scale = np.abs(jtj).max()
        assert abs(normal[c, 0] - cost) <= 1e-6 * cost
        assert np.all(np.abs(normal[c, 1:9] - jtr) <= 1e-6 * np.abs(jtr).max() + 1e-12)
        assert np.all(np.abs(normal[c, 9:] - jtj) <= 1e-6 * scale)
    # reproducible bit for bit
    _, normal2 = batch.evaluate_normal(G["poses"])
    assert np.array_equal(normal, normal2)
    # assembly into the all-reduce buffer
    size = capi.fused_size(n_nodes, batch.n_global)
    fused = torch.full((size,), float("nan"), dtype=torch.float64, device="cuda:0")
    torch.cuda.synchronize()
    batch.assemble(n_nodes, fused.data_ptr(), zero_first=True)
    ctx.synchronize()
    fused = fused.cpu().numpy()
    exp = np.zeros(size)
    iu = np.triu_indices(8)
    for c, (a, b) in enumerate(G["pairs"]):
        Hm = np.zeros((8, 8))
        Hm[iu] = normal[c, 9:]
        Hm = Hm + np.triu(Hm, 1).T
        exp[0] += normal[c, 0]
        for side, node in enumerate((a, b)):
            exp[1 + 4 * node:1 + 4 * node + 4] += normal[c, 1 + 4 * side:5 + 4 * side]
            blk = Hm[4 * side:4 * side + 4, 4 * side:4 * side + 4]
            o = 1 + 4 * n_nodes + 16 * node
            exp[o:o + 16] += blk.ravel()
        o = 1 + 20 * n_nodes + 16 * c
        exp[o:o + 16] = Hm[0:4, 4:8].ravel()
    np.testing.assert_allclose(fused, exp, rtol=1e-12, atol=1e-9)
    batch.destroy()


def test_sharded_batches_sum_to_the_unsharded_buffer(capi, ctx, small_graph):
    """Pair-sharding with N logical shards on one device: the
    per-shard fused buffers add up to the unsharded one."""
    import torch
    G = small_graph
    n_nodes, n = 4, len(G["pairs"])
    full = capi.RegistrationBatch(ctx, G["cfs"], G["pairs"])
    size = capi.fused_size(n_nodes, n)
    buf = torch.zeros(size, dtype=torch.float64, device="cuda:0")
    torch.cuda.synchronize()
    full.evaluate_normal(G["poses"], to_host=False)
    full.assemble(n_nodes, buf.data_ptr())
    ctx.synchronize()
    want = buf.cpu().numpy()
    total = np.zeros(size)
    for shard in ([0, 3], [1, 2, 4]):
        b = capi.RegistrationBatch(ctx, [G["cfs"][i] for i in shard], [G["pairs"][i] for i in shard],
                                   global_index=shard, n_global=n)
        sb = torch.zeros(size, dtype=torch.float64, device="cuda:0")
        torch.cuda.synchronize()
        b.evaluate_normal(G["poses"], to_host=False)
        b.assemble(n_nodes, sb.data_ptr())
        ctx.synchronize()
        total += sb.cpu().numpy()
        b.destroy()
    np.testing.assert_allclose(total, want, rtol=1e-12, atol=1e-9)
    full.destroy()


def test_concurrent_evaluate_from_four_threads(capi, ctx, small_graph):
    """Ceres evaluates distinct residual blocks on 4 threads (pose_graph.cpp:96): concurrent
    Evaluate calls on distinct cost functions of one context must equal the serial results."""
    from concurrent.futures import ThreadPoolExecutor
    G = small_graph

    def run(c):
        a, b = G["pairs"][c]
        cf = G["cfs"][c]
        n = cf.num_residuals()
        r, jo, je = np.zeros(n), np.zeros((n, 4)), np.zeros((n, 4))
        assert cf.Evaluate([G["poses"][a], G["poses"][b]], r, [jo, je])
        return r, jo, je

    serial = [run(c) for c in range(len(G["pairs"]))]
    with ThreadPoolExecutor(4) as ex:
        for _ in range(5):
            par = list(ex.map(run, range(len(G["pairs"]))))
            for s, p in zip(serial, par):
                assert all(np.array_equal(x, y) for x, y in zip(s, p))


def test_explicit_stream_and_timer(capi, ctx, cfg1):
    import torch
    sm, g, layer, (xyz, dist, w) = cfg1
    g.extract_voxel_points()
    cf = capi.RegistrationCostFunction(
        ctx, g, g, capi.default_config(registration_point_type=capi.POINTS_VOXELS))
    n = cf.num_residuals()
    stream = torch.cuda.Stream()
    ctx.set_stream(stream.cuda_stream)
    assert ctx.lib.vgx_ctx_get_stream(ctx.h) == stream.cuda_stream
    with torch.cuda.stream(stream):
        r = torch.zeros(n, dtype=torch.float32, device="cuda:0")
        ctx.timer_start()
        assert cf.evaluate_device_f32(np.array([0.05, 0, 0, 0.0]), np.zeros(4), r.data_ptr(), 0, 0)
        ms = ctx.timer_stop()
        total = float(r.double().abs().sum())       # ordered after the kernel by the stream
    ok0, r0, _, _ = orc.reg_evaluate(layer, xyz, dist, w, np.array([0.05, 0, 0, 0.0]), np.zeros(4), False)
    assert 0 < ms < 1000 and abs(total - np.abs(r0).sum()) < 1e-3 * np.abs(r0).sum()
    ctx.set_stream(None)
    cf.destroy()


def test_fused_pass_with_partial_overlap_culling_and_nonzero_no_correspondence_cost(capi, ctx):
    """Far-apart submaps: most 512-point chunks are culled by their bounding sphere when
    no_correspondence_cost == 0 (exact), and nothing is culled when it is != 0."""
    sdf = synth.sphere_ground_sdf((0.3, -0.2, 0.1), 1.5, -1.0)
    ref = synth.make_submap(sdf, 0.1, 16, (-3, -3, -2), (6, 6, 4), trunc=0.3, esdf_max=0.8)
    read = synth.make_submap(sdf, 0.1, 16, (0, -1, -1), (3, 3, 3), trunc=0.3, esdf_max=0.8,
                             pose=(0.4, 0.1, -0.2, 0.3))
    g_ref, g_read = H.gpu_submap(capi, ctx, ref, 0), H.gpu_submap(capi, ctx, read, 1)
    n = g_ref.extract_voxel_points()
    xyz, dist, w = H.oracle_points(ref)
    layer = H.oracle_layer(read)
    poses = np.array([[0.0, 0.0, 0.0, 0.0], [0.45, 0.05, -0.22, 0.27]])
    for ncc in (0.0, 0.37):
        cfg = capi.default_config(registration_point_type=capi.POINTS_VOXELS, no_correspondence_cost=ncc)
        cf = capi.RegistrationCostFunction(ctx, g_ref, g_read, cfg)
        batch = capi.RegistrationBatch(ctx, [cf], [(0, 1)])
        status, normal = batch.evaluate_normal(poses)
        ok, cost, jtr, jtj = orc.reg_evaluate_normal(layer, xyz, dist, w, poses[0], poses[1],
                                                     no_correspondence_cost=ncc)
        ok2, r0, jo0, _ = orc.reg_evaluate(layer, xyz, dist, w, poses[0], poses[1],
                                           no_correspondence_cost=ncc)
        assert ok and 0.05 < np.any(jo0 != 0, axis=1).mean() < 0.6     # mostly outside
        assert abs(normal[0, 0] - cost) <= 1e-6 * cost
        assert np.all(np.abs(normal[0, 1:9] - jtr) <= 1e-6 * np.abs(jtr).max())
        assert np.all(np.abs(normal[0, 9:] - jtj) <= 1e-6 * np.abs(jtj).max())
        # the materialising pass culls whole 1024-point tiles the same way (rows of zeros written
        # without reading the points): batched f32 launch and the f64 drop-in Evaluate, row for row
        import torch
        r = _torch_buf(n, torch.float32)
        jo, je = _torch_buf(4 * n, torch.float32), _torch_buf(4 * n, torch.float32)
        torch.cuda.synchronize()
        assert np.all(batch.evaluate_points(poses, r.data_ptr(), jo.data_ptr(), je.data_ptr()) == 0)
        ctx.synchronize()
        assert batch.launch_order(points_pass=True) == 0 and batch.launch_order(points_pass=False) == 0
        H.assert_parity(r.cpu().numpy(), r0, "culled tiles: residual")
        H.assert_parity(jo.cpu().numpy().reshape(n, 4), jo0, "culled tiles: jac_ref")
        dead_rows = ~np.any(jo0 != 0, axis=1)
        if ncc == 0.0:
            assert np.all(r.cpu().numpy()[dead_rows] == 0) and np.all(je.cpu().numpy().reshape(n, 4)[dead_rows] == 0)
        okg, rg, jog, jeg = _gpu_eval(cf, poses[0], poses[1])
        assert okg
        H.assert_parity(rg, r0, "culled tiles, drop-in: residual")
        H.assert_parity(jog, jo0, "culled tiles, drop-in: jac_ref")
        batch.destroy()
        cf.destroy()
    g_ref.destroy()
    g_read.destroy()


def test_live_counts_per_constraint_sum_to_the_batch_count(capi, ctx, small_graph):
    G = small_graph
    batch = capi.RegistrationBatch(ctx, G["cfs"], G["pairs"])
    each = batch.count_live_each(G["poses"])
    total, distinct = batch.count_live(G["poses"], unique=True)
    assert each.sum() == total and 0 < distinct <= total
    assert np.all(each <= [cf.num_residuals() for cf in G["cfs"]])
    far = G["poses"].copy()
    far[:, 0] += 1e3 * np.arange(len(far))
    assert batch.count_live_each(far).sum() == 0 and batch.launch_order(points_pass=True) == -1
    batch.destroy()


def test_compressed_blocks_reproduce_the_normal_equations(capi, ctx, small_graph):
    """vgx_reg_compress_normal: 9 residuals per constraint with the same J^T J, J^T r, r^T r
    (what voxgraph_amd/cpp/gpu_registration_batch.h hands to Ceres)."""
    G = small_graph
    batch = capi.RegistrationBatch(ctx, G["cfs"], G["pairs"])
    _, normal = batch.evaluate_normal(G["poses"])
    iu = np.triu_indices(8)
    for nb in normal:
        r, J = capi.compress_normal(nb)
        Hm = np.zeros((8, 8))
        Hm[iu] = nb[9:]
        Hm = Hm + np.triu(Hm, 1).T
        scale = np.abs(Hm).max()
        # The 45 numbers are sums of f32-accumulated partial sums (lean fused kernel): [J r]^T [J r]
        # is positive semi-definite only up to their rounding (~1e-8 relative), and the compression
        # clamps the slightly negative eigenvalues of weakly constrained directions to zero.
        assert np.abs(J.T @ J - Hm).max() <= 1e-7 * scale
        assert np.abs(J.T @ r - nb[1:9]).max() <= 1e-7 * max(np.abs(nb[1:9]).max(), 1e-300) + 1e-9 * scale
        assert abs(r @ r - nb[0]) <= 1e-7 * nb[0]
    batch.destroy()


def test_randomised_configurations_match_oracle(capi, ctx):
    """25 seeded random set-ups: sparse random block sets (negative indices), both vps, voxel
    sizes from 2 cm to 50 cm, poses hundreds of metres from the origin and yaw across the
    +-pi wrap, random weights, unobserved voxels, both distance layers, both output forms."""
    worst = 0.0
    for seed in range(25):
        rng = np.random.default_rng(1000 + seed)
        vps = 16 if seed % 3 else 8
        vs = float(rng.choice([0.02, 0.05, 0.1, 0.2, 0.5]))
        bdim = tuple(int(x) for x in rng.integers(2, 5, 3))
        bmin = tuple(int(x) for x in rng.integers(-6, 3, 3))
        extent = np.array(bdim) * vps * vs
        centre = (np.array(bmin) * vps * vs) + extent * rng.uniform(0.3, 0.7, 3)
        sdf = synth.union_sdf(synth.sphere_sdf(centre, 0.3 * extent.min()),
                              synth.plane_sdf((0.1, -0.2, 0.97), float(np.array([0.1, -0.2, 0.97]) @ centre) - 0.2 * extent.min()))
        trunc = 3 * vs
        sm = synth.make_submap(sdf, vs, vps, bmin, bdim, trunc=trunc, esdf_max=6 * vs, noise=0.02, seed=seed)
        keep = rng.random(sm.n_blocks) < 0.8                      # random missing blocks
        keep[0] = True
        for name in ("block_index", "tsdf_distance", "tsdf_weight", "esdf_distance", "esdf_observed"):
            setattr(sm, name, np.ascontiguousarray(getattr(sm, name)[keep]))
        sm.esdf_observed[rng.random(sm.esdf_observed.shape) < 0.01] = 0    # unobserved speckles
        use_esdf = bool(seed % 2)
        g = H.gpu_submap(capi, ctx, sm, seed)
        xyz, dist, w = H.oracle_points(sm, use_esdf=use_esdf, max_d=1.0 * trunc)
        if len(w) < 50:
            g.destroy()
            continue
        w = (w * rng.uniform(0.2, 1.0, len(w))).astype(F)
        flags = capi.POINTS_SORT_MORTON if seed % 4 == 0 else capi.POINTS_KEEP_ORDER
        g.set_points(capi.POINTS_VOXELS, xyz, dist, w, flags)
        order = g.point_order(capi.POINTS_VOXELS)
        layer = H.oracle_layer(sm, use_esdf=use_esdf)
        cfg = capi.default_config(registration_point_type=capi.POINTS_VOXELS,
                                  use_esdf_distance=int(use_esdf),
                                  no_correspondence_cost=float(rng.choice([0.0, 0.0, 0.3])))
        cf = capi.RegistrationCostFunction(ctx, g, g, cfg)
        base = np.r_[rng.uniform(-400, 400, 2), rng.uniform(-20, 20), rng.uniform(-3.14, 3.14)]
        ref_pose = base
        d_yaw = rng.uniform(-0.3, 0.3)
        read_pose = base + np.r_[rng.normal(0, 2 * vs, 3), d_yaw]
        if seed % 5 == 0:
            ref_pose = ref_pose.copy(); ref_pose[3] = 3.13; read_pose[3] = -3.12   # across the wrap
        ok, r, jo, je = _gpu_eval(cf, ref_pose, read_pose)
        ok0, r0, jo0, je0 = orc.reg_evaluate(layer, xyz, dist, w, ref_pose, read_pose,
                                             no_correspondence_cost=cfg.no_correspondence_cost)
        assert ok == ok0
        if not ok:
            continue
        worst = max(worst, H.assert_parity(r, r0[order], f"residual seed {seed}"),
                    H.assert_parity(jo, jo0[order], f"jac_ref seed {seed}"),
                    H.assert_parity(je, je0[order], f"jac_read seed {seed}"))
        batch = capi.RegistrationBatch(ctx, [cf], [(0, 1)])
        _, normal = batch.evaluate_normal(np.vstack([ref_pose, read_pose]))
        okn, cost, jtr, jtj = orc.reg_evaluate_normal(layer, xyz, dist, w, ref_pose, read_pose,
                                                      no_correspondence_cost=cfg.no_correspondence_cost)
        assert abs(normal[0, 0] - cost) <= 1e-6 * max(cost, 1e-12)
        assert np.all(np.abs(normal[0, 9:] - jtj) <= 1e-6 * max(np.abs(jtj).max(), 1e-12))
        for o in (batch, cf, g):
            o.destroy()
    print("randomised configurations: worst relative error", worst)


# ------------------------------------------------ reference-source golden -----
@pytest.mark.parametrize("case,use_esdf,no_corr,ratio", [
    ("esdf_all", True, 0.0, -1.0), ("tsdf_nocorr", False, 0.7, -1.0), ("esdf_sampled", True, 0.0, 0.05)])
def test_hip_path_matches_the_reference_source_golden(capi, ctx, golden_dir, case, use_esdf, no_corr, ratio):
    """tests/golden/ref_reg_config1.npz holds outputs of the REFERENCE's own
    registration_cost_function.cpp (compiled against oracle/ref_shims, see
    tests/golden/make_ref_golden.py).  The HIP path, through the C ABI with device-side point
    extraction, must match them to the contract's 1e-4 -- and does so bit for bit."""
    import hashlib
    import os
    gold = np.load(os.path.join(golden_dir, "ref_reg_config1.npz"))
    ref, read = synth.config1_pair(seed=0, asymmetric=True)
    g_ref, g_read = H.gpu_submap(capi, ctx, ref, 30), H.gpu_submap(capi, ctx, read, 31)
    n_pts = g_ref.extract_voxel_points(1.0, 0.3, use_esdf)
    cfg = capi.default_config(registration_point_type=capi.POINTS_VOXELS, sampling_ratio=ratio,
                              no_correspondence_cost=no_corr, use_esdf_distance=int(use_esdf))
    cf = capi.RegistrationCostFunction(ctx, g_ref, g_read, cfg)
    n = cf.num_residuals()
    assert n == int(gold[f"{case}_num_residuals"]) and n <= n_pts
    base, stride = gold["base_pose"], int(gold["stride"])
    exact = 0
    for k, pert in enumerate(gold["perturbations"]):    # successive calls continue the sampler stream
        ok, r, jo, je = _gpu_eval(cf, base, base + pert)
        assert ok
        key = f"{case}_{k}"
        H.assert_parity(r[::stride], gold[key + "_r"], key + " residual")
        H.assert_parity(jo[::stride], gold[key + "_jref"], key + " jac_ref")
        H.assert_parity(je[::stride], gold[key + "_jread"], key + " jac_read")
        assert int((np.abs(jo).sum(1) > 0).sum()) == int(gold[key + "_corr"])
        assert abs(0.5 * (r * r).sum() - float(gold[key + "_cost"])) <= 1e-9 * float(gold[key + "_cost"])
        # `+ 0.0`: -0.0 and +0.0 are the same value (make_ref_golden.py digest())
        sha = [hashlib.sha256(np.ascontiguousarray(x + 0.0).tobytes()).hexdigest() for x in (r, jo, je)]
        exact += int(sha == list(gold[key + "_sha"]))
    print(f"{case}: {exact}/{len(gold['perturbations'])} evaluations value-identical (every f64 equal) to the reference source")
    assert exact == len(gold["perturbations"])
    for o in (cf, g_ref, g_read):
        o.destroy()


def test_a_submap_outlives_the_cost_functions_built_on_it(capi, ctx):
    """Lifetimes as the reference's: RegistrationCostFunction holds VoxgraphSubmap::ConstPtr to both submaps and the
    ceres::Problem owns its cost functions, so vgx_submap_destroy on a submap cost functions were built on -- what
    GpuSubmapRegistry does to a stale upload -- and vgx_reg_destroy on a cost function a batch lists are DEFERRED to the
    last user's destruction (include/voxgraph_amd.h): evaluations in between are the ones from before, bit for bit."""
    import torch
    sm, _ = synth.config1_pair()
    a, b = H.gpu_submap(capi, ctx, sm, 31), H.gpu_submap(capi, ctx, sm, 32)
    a.extract_voxel_points()
    cf = capi.RegistrationCostFunction(ctx, a, b, capi.default_config(registration_point_type=capi.POINTS_VOXELS))
    cf2 = capi.RegistrationCostFunction(ctx, a, a, capi.default_config(registration_point_type=capi.POINTS_VOXELS))
    ref_pose, read_pose = np.array([0.02, -0.01, 0.03, 0.01]), np.array([0.07, 0.03, -0.02, 0.04])
    ok, r0, jo0, je0 = _gpu_eval(cf, ref_pose, read_pose)
    assert ok
    batch = capi.RegistrationBatch(ctx, [cf, cf2], [[0, 1], [0, 1]])
    _, n0 = batch.evaluate_normal(np.stack([ref_pose, read_pose]))
    # the owners let go in the "wrong" order: submaps first, then the cost functions, the batch last
    a.destroy()
    b.destroy()
    junk = [torch.full((1 << 22,), 7.0, device="cuda") for _ in range(8)]  # whatever was freed would be handed out again
    torch.cuda.synchronize()
    ok, r1, jo1, je1 = _gpu_eval(cf, ref_pose, read_pose)
    assert ok and np.array_equal(r0, r1) and np.array_equal(jo0, jo1) and np.array_equal(je0, je1)
    cf.destroy()
    cf2.destroy()
    _, n1 = batch.evaluate_normal(np.stack([ref_pose, read_pose]))
    assert np.array_equal(n0, n1) and np.abs(n0).sum() > 0
    batch.destroy()   # -> the two cost functions -> the two submaps
    del junk
