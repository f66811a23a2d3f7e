"""Is the registration bias of the config-2 stand-in DATA or IMPLEMENTATION?  

The session of harness/pipeline.py, cut to a few submaps, is pushed through two complete chains
that share nothing but the scans and the solver:

  GPU chain     vgx_tsdf_integrate_device -> vgx_submap_from_tsdf_layer -> vgx_submap_generate_esdf
                -> vgx_submap_extract_{isosurface,voxel}_points -> fused REG pass -> harness/lm.py
  oracle chain  oracle/tsdf_oracle.c -> oracle/esdf_oracle.c -> oracle/iso_oracle.c | reg_oracle.c's
                findRelevantVoxelIndices -> oracle/reg_oracle.c normal equations -> harness/lm.py
                (the restatements that tests/test_ref_pin.py and tests/test_ref_submap_pin.py pin to
                the reference's own registration_cost_function.cpp / voxgraph_submap.cpp; where
                oracle/_ref is present the reference's cost function re-evaluates the oracle chain's
                end state as a cross-check)

in the reference's DEFAULT mode (use_esdf_distance = true, registration_cost_function.h:35), started
from the ground truth ("how far does the optimum sit from the truth") and from the drifted odometry.
If both chains drift alike, the bias is a property of the data (synthetic street canyon + voxblox's
TSDF/ESDF semantics), not of the kernels.  Also compared: the ESDF the GPU makes from the ORACLE's
TSDF against the oracle's ESDF of the same TSDF (the one deterministic producer in the chain).

Measurement / test infrastructure: imports oracle/ (tests and bench only)."""
import numpy as np

from . import lm
from .backends import GpuBackend, OracleBackend
from .pipeline import _inv_compose, session_sensor_poses


def _between(pa, pb):
    c, s_ = np.cos(pa[3]), np.sin(pa[3])
    d = pb[:3] - pa[:3]
    return np.array([c * d[0] + s_ * d[1], -s_ * d[0] + c * d[1], d[2], lm.normalize_angle(pb[3] - pa[3])])


def _compose(pose, delta):
    c, s_ = np.cos(pose[3]), np.sin(pose[3])
    return np.array([pose[0] + c * delta[0] - s_ * delta[1], pose[1] + s_ * delta[0] + c * delta[1],
                     pose[2] + delta[2], lm.normalize_angle(pose[3] + delta[3])])


def run(capi, ctx, torch, n_submaps=6, scans_per_submap=30, n_az=512, n_el=32, voxel_size=0.2,
        full_session_submaps=30, seed=1, drift_sigma=(0.12, 0.008), use_esdf_distance=True,
        isosurface_points=True, deterministic_tsdf=True):
    """deterministic_tsdf: the GPU chain integrates in the reproducible mode (vgx_tsdf_config.deterministic),
    i.e. in the oracle's own single-thread order -- its TSDF layers are then the oracle's bit for bit and
    whatever separates the two chains comes from the later stages alone."""
    from oracle import pyoracle as orc
    rng = np.random.default_rng(seed)
    vps = 16
    # the first n_submaps of the full session's trajectory (same spacing as the 30-submap lap)
    poses_all = session_sensor_poses(full_session_submaps, scans_per_submap)
    sensor_poses = poses_all[:n_submaps * scans_per_submap]
    el_span = np.deg2rad(33.2)
    gcfg, ocfg = capi.voxgraph_tsdf_config(deterministic=int(bool(deterministic_tsdf))), orc.voxgraph_tsdf_config()
    pts = torch.empty((n_az * n_el, 3), dtype=torch.float32, device="cuda")
    g_sub, true_poses = [], []
    o_layers, o_points, o_tsdf = [], [], []
    esdf_cmp, tsdf_cmp = [], []
    for m in range(n_submaps):
        first = m * scans_per_submap
        P = sensor_poses[first].copy()
        P[2] = 0.0
        true_poses.append(P)
        glayer = capi.TsdfLayer(ctx, voxel_size, vps)
        ginteg = capi.FastTsdfIntegrator(ctx, gcfg, glayer)
        olayer = orc.TsdfLayer(voxel_size, vps)
        ointeg = orc.FastTsdfIntegrator(ocfg, olayer)
        for j in range(first, first + scans_per_submap):
            capi.synth_city_scan(ctx, sensor_poses[j], n_az, n_el, el_span, 40.0, 2, pts.data_ptr())
            ctx.synchronize()
            T = _inv_compose(P, sensor_poses[j])
            ginteg.integrate_device(T, pts.data_ptr(), None, n_az * n_el)
            ointeg.integratePointCloud(T, pts.cpu().numpy())
        # ---- GPU finishSubmap ----
        sm = capi.Submap.from_tsdf_layer(ctx, glayer, m)
        sm.generate_esdf()
        sm.extract_voxel_points(1.0, 0.3, use_esdf_distance)
        sm.extract_isosurface_points(1.0)
        g_sub.append(sm)
        # ---- oracle finishSubmap ----
        bi, td, tw, _ = olayer.download()
        ed, eo, _ = orc.esdf_from_tsdf(voxel_size, vps, bi, td, tw)
        if isosurface_points:
            pxyz, pd, pw = orc.isosurface_points(voxel_size, vps, bi, td, tw, 1.0)
        else:
            pxyz, pd, pw = orc.find_relevant_voxels(voxel_size, vps, bi, td, tw, ed if use_esdf_distance else None)
        o_points.append((pxyz, pd, pw))
        o_layers.append(orc.Layer(voxel_size, vps, bi, ed, eo) if use_esdf_distance
                        else orc.Layer(voxel_size, vps, bi, td, (tw > 0).astype(np.uint8)))
        o_tsdf.append((bi, td, tw, ed, eo))
        # ---- the deterministic producer, same input on both sides: ESDF from the ORACLE's TSDF ----
        probe = capi.Submap(ctx, 1000 + m, voxel_size, vps, bi, td, tw, None, None)
        probe.generate_esdf()
        _, _, ged, geo = probe.download_layers(vps)
        probe.destroy()
        obs = eo.astype(bool)
        diff = np.abs(ged - ed)[obs]
        esdf_cmp.append(dict(observed_equal=bool(np.array_equal(geo, eo)), max=float(diff.max()),
                             p99=float(np.percentile(diff, 99)), mean=float(diff.mean()),
                             n_observed=int(obs.sum()),
                             gpu_never_above_oracle=bool(np.all(np.abs(ged[obs]) <= np.abs(ed[obs]) + 1e-6))))
        # ---- TSDF, GPU's own against the oracle's (legal orders differ on dense scans) ----
        gbi, gtd, gtw, _ = glayer.download()
        gd = {tuple(b): k for k, b in enumerate(gbi)}
        common = [(gd[tuple(b)], k) for k, b in enumerate(bi) if tuple(b) in gd]
        ia, ib = np.array([c[0] for c in common]), np.array([c[1] for c in common])
        both = (gtw[ia] > 0) & (tw[ib] > 0)
        dd = np.abs(gtd[ia][both] - td[ib][both])
        tsdf_cmp.append(dict(blocks_gpu=int(len(gbi)), blocks_oracle=int(len(bi)), blocks_common=len(common),
                             observed_both=int(both.sum()), p50=float(np.percentile(dd, 50)),
                             p99=float(np.percentile(dd, 99)), max=float(dd.max()),
                             bit_identical=bool(np.array_equal(gbi, bi) and np.array_equal(gtd.view(np.uint32), td.view(np.uint32))
                                                and np.array_equal(gtw.view(np.uint32), tw.view(np.uint32)))))
        for o in (ginteg, glayer):
            o.destroy()
    true_poses = np.array(true_poses)
    # the constraint list: the device's overlap test at the true poses, both directions for
    # isosurface points (pose_graph.cpp:62-71); the SAME list drives both chains
    pairs = capi.find_overlapping_pairs(ctx, g_sub, true_poses)
    if isosurface_points:
        pairs = pairs + [(b, a) for a, b in pairs]
    ptype = capi.POINTS_ISOSURFACE if isosurface_points else capi.POINTS_VOXELS
    rcfg = capi.default_config(registration_point_type=ptype, use_esdf_distance=int(use_esdf_distance))
    cfs = [capi.RegistrationCostFunction(ctx, g_sub[a], g_sub[b], rcfg) for a, b in pairs]
    batch = capi.RegistrationBatch(ctx, cfs, pairs)
    gpu_backend = GpuBackend(capi, ctx, batch, n_submaps)
    orc_backend = OracleBackend(o_layers, o_points, pairs, n_submaps, threads=8)
    # drifted odometry, as harness/pipeline.py draws it
    odom, drifted = [], true_poses[:1].copy()
    for m in range(1, n_submaps):
        delta = _between(true_poses[m - 1], true_poses[m]) + np.r_[
            rng.normal(0, drift_sigma[0], 2), 0.2 * rng.normal(0, drift_sigma[0]), rng.normal(0, drift_sigma[1])]
        odom.append(delta)
        drifted = np.vstack([drifted, _compose(drifted[m - 1], delta)])
    info = [1.0, 1.0, 2500.0, 2500.0]
    edges = [lm.RelativePoseEdge(k, k + 1, odom[k][:3], odom[k][3], info) for k in range(n_submaps - 1)]
    kw = dict(parameter_tolerance=1e-8, max_seconds=1e9)

    def rmse(p):
        return float(np.sqrt(((p[:, :2] - true_poses[:, :2]) ** 2).sum(1).mean()))
    out = {"submaps": n_submaps, "scans_per_submap": scans_per_submap, "points_per_scan": n_az * n_el,
           "tsdf_mode": "reproducible" if deterministic_tsdf else "racing",
           "mode": ("kIsosurfacePoints mirrored, " if isosurface_points else "kVoxels, ") +
                   ("ESDF distance" if use_esdf_distance else "TSDF distance"),
           "constraints": len(pairs), "xy_rmse_m_odometry_only": rmse(drifted),
           "esdf_gpu_vs_oracle_same_tsdf": esdf_cmp, "tsdf_gpu_vs_oracle": tsdf_cmp,
           "points_per_submap_gpu": [int(s.num_points(ptype)) for s in g_sub],
           "points_per_submap_oracle": [int(len(p[2])) for p in o_points]}
    same_points = []
    for m, sm in enumerate(g_sub):
        gx, gd_, gw_ = sm.download_points(ptype)
        ox, od_, ow_ = o_points[m]
        ox, od_, ow_ = (np.ascontiguousarray(a, np.float32) for a in (ox, od_, ow_))
        same_points.append(bool(gx.shape == ox.shape and np.array_equal(gx.view(np.uint32), ox.view(np.uint32))
                                and np.array_equal(gd_.view(np.uint32), od_.view(np.uint32))
                                and np.array_equal(gw_.view(np.uint32), ow_.view(np.uint32))))
    out["registration_points_bit_identical"] = same_points
    ends = {}
    for start_name, start in (("from_truth", true_poses), ("from_drift", drifted)):
        for chain, backend in (("gpu", gpu_backend), ("oracle", orc_backend)):
            x, summ = lm.solve(lm.Problem(backend, n_submaps, pairs, edges), start, **kw)
            ends[(start_name, chain)] = x
            out[f"{start_name}_{chain}"] = {"xy_rmse_m": rmse(x), "iterations": summ["iterations"],
                                            "termination": summ["termination"], "final_cost": summ["final_cost"]}
        d = ends[(start_name, "gpu")] - ends[(start_name, "oracle")]
        out[f"{start_name}_end_pose_difference"] = {"xy_max_m": float(np.abs(d[:, :2]).max()),
                                                    "yaw_max_rad": float(np.abs(lm.normalize_angle(d[:, 3])).max())}
    # the reference's own cost function on the oracle chain's end state (when oracle/_ref travelled here)
    try:
        from oracle import ref_reg
        if ref_reg.available():
            x = ends[("from_truth", "oracle")]
            refs = []
            for m, (bi, td, tw, ed, eo) in enumerate(o_tsdf):
                R = ref_reg.Submap(m, x[m], voxel_size, vps, bi, td, tw, ed, eo)
                R.set_points(ref_reg.POINTS_ISOSURFACE if isosurface_points else ref_reg.POINTS_VOXELS, *o_points[m])
                refs.append(R)
            ref_cost = orc_cost = 0.0
            for a, b in pairs:
                cf = ref_reg.RegistrationCostFunction(
                    refs[a], refs[b], point_type=ref_reg.POINTS_ISOSURFACE if isosurface_points else ref_reg.POINTS_VOXELS,
                    use_esdf_distance=use_esdf_distance)
                ok, r, _, _ = cf.Evaluate(x[a], x[b], want_jac=False)
                ok2, r2, _, _ = orc.reg_evaluate(o_layers[b], *o_points[a], x[a], x[b], want_jac=False)
                assert ok and ok2
                ref_cost += float(r @ r)
                orc_cost += float(r2 @ r2)
            out["reference_source_cost_at_oracle_end_state"] = {"reference": ref_cost, "oracle": orc_cost,
                                                               "equal": bool(ref_cost == orc_cost)}
    except Exception as e:                                      # the cross-check must never sink the comparison
        out["reference_source_cost_at_oracle_end_state"] = {"error": repr(e)}
    for o in [batch] + cfs + g_sub:
        o.destroy()
    return out
