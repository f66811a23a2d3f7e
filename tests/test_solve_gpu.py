"""End-to-end parity: the same harness LM solver (harness/lm.py, stand-in for
ceres::Solve) driven by the GPU library and by the CPU oracle must end at the same
submap poses -- north_star: within 1 mm / 0.01 deg."""
import numpy as np
import pytest

from harness import lm
from harness.backends import GpuBackend, OracleBackend
from oracle import synth
from tests import helpers as H

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def capi():
    from voxgraph_amd import capi
    capi.load()
    return capi


@pytest.fixture(scope="module")
def ctx(capi):
    import torch
    c = capi.Context(0)
    c.set_stream(torch.cuda.current_stream().cuda_stream)
    yield c
    c.close()


def _pose_err(x, y):
    dt = np.abs(x[:, :3] - y[:, :3]).max()
    dyaw = np.abs(lm.normalize_angle(x[:, 3] - y[:, 3])).max()
    return dt, np.rad2deg(dyaw)


def test_known_answer_solve_on_gpu(capi, ctx):
    """Test-bench design on the GPU path: duplicated (asymmetric) config-1 submap,
    perturbed on the reference's grid, must return to the unperturbed pose."""
    sm, _ = synth.config1_pair(asymmetric=True)
    g = H.gpu_submap(capi, ctx, sm)
    g.extract_voxel_points()
    cfg = capi.default_config(registration_point_type=capi.POINTS_VOXELS)
    cf = capi.RegistrationCostFunction(ctx, g, g, cfg)
    batch = capi.RegistrationBatch(ctx, [cf], [(0, 1)])
    backend = GpuBackend(capi, ctx, batch, 2)
    grid = H.test_bench_grid(sm.voxel_size)
    for pert in grid[::7]:
        prob = lm.Problem(backend, 2, [(0, 1)])
        x, s = lm.solve(prob, np.array([[0.0, 0, 0, 0], pert]), parameter_tolerance=1e-9,
                        function_tolerance=1e-14, max_iterations=60, max_seconds=60)
        dt, dyaw = _pose_err(x[1:], np.zeros((1, 4)))
        assert dt < 1e-3 and dyaw < 0.01, (pert, x[1], s)
    for o in (batch, cf, g):
        o.destroy()


def test_final_poses_match_cpu_oracle_backend(capi, ctx):
    sdf = synth.union_sdf(synth.sphere_ground_sdf((1.6, 1.6, 1.2), 1.0, 0.35),
                          synth.sphere_sdf((0.6, 2.4, 0.8), 0.5))
    true = np.array([(0, 0, 0, 0), (0.8, 0.1, 0.0, 0.1), (0.1, 0.9, 0.05, -0.15), (0.9, 0.8, 0.0, 0.2)])
    sms, gs, layers, pts = [], [], [], []
    for i, p in enumerate(true):
        sm = synth.make_submap(sdf, 0.1, 16, (0, 0, 0), (2, 2, 2), 0.3, p, 1.0, drop_empty_blocks=True)
        g = H.gpu_submap(capi, ctx, sm, i)
        g.extract_voxel_points()
        sms.append(sm), gs.append(g), layers.append(H.oracle_layer(sm)), pts.append(H.oracle_points(sm))
    pairs = [(0, 1), (0, 2), (1, 3), (2, 3), (1, 2), (0, 3)]
    cfg = capi.default_config(registration_point_type=capi.POINTS_VOXELS)
    cfs = [capi.RegistrationCostFunction(ctx, gs[a], gs[b], cfg) for a, b in pairs]
    batch = capi.RegistrationBatch(ctx, cfs, pairs)
    rng = np.random.default_rng(4)
    poses0 = true + np.concatenate([rng.normal(0, 0.06, (4, 3)), rng.normal(0, 0.04, (4, 1))], 1)
    poses0[0] = true[0]
    info = [1.0, 1.0, 2500.0, 2500.0]            # voxgraph_mapper.yaml:41-47
    edges = [lm.RelativePoseEdge.from_poses(k, k + 1, poses0[k], poses0[k + 1], info) for k in range(3)]
    gpu = lm.Problem(GpuBackend(capi, ctx, batch, 4), 4, pairs, edges)
    cpu = lm.Problem(OracleBackend(layers, pts, pairs, 4), 4, pairs, edges)
    kw = dict(parameter_tolerance=1e-8, function_tolerance=1e-12, max_iterations=40, max_seconds=120)
    xg, sg = lm.solve(gpu, poses0, **kw)
    xc, sc = lm.solve(cpu, poses0, **kw)
    dt, dyaw = _pose_err(xg, xc)
    print("gpu", sg, "\ncpu", sc, "\nGPU vs CPU-oracle final poses: dt", dt, "m, dyaw", dyaw, "deg")
    assert dt < 1e-3 and dyaw < 0.01
    assert sg["final_cost"] < 0.5 * sg["initial_cost"]
    # also with the reference's stop rule (parameter_tolerance 3e-3, pose_graph.cpp:93)
    xg2, _ = lm.solve(lm.Problem(GpuBackend(capi, ctx, batch, 4), 4, pairs, edges), poses0)
    xc2, _ = lm.solve(lm.Problem(OracleBackend(layers, pts, pairs, 4), 4, pairs, edges), poses0)
    dt2, dyaw2 = _pose_err(xg2, xc2)
    assert dt2 < 1e-3 and dyaw2 < 0.01
    for o in [batch] + cfs + gs:
        o.destroy()
