#!/usr/bin/env python3
"""bench.py -- REG hot path on BASELINE config 3 (200 submaps @ 256^3, pair-sharded).

One "step" = `--inner` (default 10: a converged solve of this graph takes 9) consecutive
passes of the registration hot path over the whole pose graph's registration
constraints: in every pass every constraint's residuals and both Jacobians are
evaluated (materialising f32 form, 88 B/evaluation) by ONE batched
launch per rank.  Inputs (sampling grids, registration points) are resident in
HBM before the timed region; the only per-pass host->device traffic is the 64-B
pose pack per constraint.  `value` counts evaluations, so it does not depend on
`--inner`; the driver's 20 steps then time >= 1 s instead of 0.1 s.

A second REG workload on the same submaps measures the gather-dominated regime the
88 B contract figure describes (`roofline_full_overlap`): every submap registered against
duplicates of itself perturbed by N(0, 0.3 m) / N(0, 0.05 rad) -- the reference's own
test-bench design (registration_test_bench.cpp:178-185) -- so ~100 % of the evaluations
interpolate.  Constraints are ordered so that consecutive ones never share a submap.

N > 1: the constraint list is sharded across ranks (greedy LPT by residual
count), submaps are replicated, no data-path collective is needed
for this materialising pass; the fixed graph makes this STRONG scaling.  The
fused pass + RCCL all-reduce of the normal-equation buffer (what a solver
iteration needs) is timed after the headline region and reported under
"fused" (it does not contribute to `value`).

  python bench.py --gpus 1 --steps 10 --warmup 3
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
      --master-port P bench.py --gpus N --steps K --warmup W
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
# the host driver only supports dmabuf IPC: without this RCCL's peer mapping fails with
# "hipIpcGetMemHandle: invalid argument" (already exported on the GPU boxes; kept for safety)
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")

HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
BYTES_PER_EVAL = 88            # SURVEY.md 8d contract figure (materialising, f32 outputs)
BYTES_NO_CORR = 56             # an evaluation that finds no reading block: 20 B in, 36 B out
BYTES_OUT, BYTES_POINT, BYTES_NEIGHBOURS = 36, 20, 32   # the three parts of the 88 B
BYTES_PER_EVAL_FUSED = 52      # fused form: 20 B point + 32 B neighbours, nothing written per point
BYTES_NO_CORR_FUSED = 20       # a point the fused pass loads but that finds no reading block
PROFILE_TRAFFIC = {}           # profiles/hbm_traffic.json: PMC bytes per launch of the workloads below


def build_graph(args):
    """Config 3: grid of submaps over the analytic city scene, constraints between
    submaps whose cubes overlap."""
    gw, gh = args.grid
    rng = np.random.default_rng(args.seed)
    dims = np.array(args.block_dims, np.int32)
    extent = dims * 16 * args.voxel_size
    sx, sy = extent[0] * 0.5, extent[1] / 3.0       # 50 % overlap in x, 2/3 in y
    true_poses, ids = [], {}
    for j in range(gh):
        for i in range(gw):
            ids[(i, j)] = len(true_poses)
            true_poses.append([i * sx, j * sy, 0.0, rng.uniform(-0.1, 0.1)])
    true_poses = np.array(true_poses)
    pairs = []
    for j in range(gh):
        for i in range(gw):
            for di, dj in ((1, 0), (0, 1), (1, 1), (-1, 1), (0, 2), (1, 2), (-1, 2)):
                k = (i + di, j + dj)
                if k in ids:
                    pairs.append((ids[(i, j)], ids[k]))
    # initial guess = truth + drift: N(0, 0.3 m), N(0, 0.05 rad); submap 0 fixed
    poses = true_poses + np.concatenate(
        [rng.normal(0, args.pose_sigma, (len(true_poses), 3)),
         rng.normal(0, args.yaw_sigma, (len(true_poses), 1))], axis=1)
    poses[0] = true_poses[0]
    return true_poses, poses, np.array(pairs, np.int32)


def lpt_shards(weights, n):
    """Greedy longest-processing-time partition of constraints onto n ranks: the library's own
    placement (vgx_lpt_shards), the one the in-process multi-GPU component uses."""
    from voxgraph_amd import capi
    shard_of = capi.lpt_shards(weights, n)
    return [[int(c) for c in np.nonzero(shard_of == r)[0]] for r in range(n)]


def cpu_baseline(capi, ctx, args, true_poses, poses, pairs, seconds):
    """The CPU oracle ("port") timed on this host on ONE constraint of the same
    workload, replicated over all host cores (one constraint per task, the
    reference's parallelism axis, pose_graph.cpp:96)."""
    from concurrent.futures import ThreadPoolExecutor
    from oracle import pyoracle as orc
    a, b = int(pairs[0][0]), int(pairs[0][1])
    subs = {}
    for k in (a, b):
        sm = capi.Submap.synth_city(ctx, k, args.voxel_size, 16, args.block_min, args.block_dims,
                                    args.truncation, args.esdf_max, 10.0, true_poses[k], args.seed)
        td, tw, ed, eo = sm.download_layers(16)
        subs[k] = (sm.block_index(), td, tw, ed, eo)
        sm.destroy()
    bi, td, tw, ed, eo = subs[a]
    xyz, dist, w = orc.find_relevant_voxels(args.voxel_size, 16, bi, td, tw, ed)
    bi, td, tw, ed, eo = subs[b]
    layer = orc.Layer(args.voxel_size, 16, bi, ed, eo)
    cores = os.cpu_count() or 1
    n = len(w)

    def task(_):
        ok, r, jo, je = orc.reg_evaluate(layer, xyz, dist, w, poses[a], poses[b])
        return n

    task(0)                                   # page everything in
    done, t0 = 0, time.perf_counter()
    with ThreadPoolExecutor(cores) as ex:
        while time.perf_counter() - t0 < seconds:
            done += sum(ex.map(task, range(cores)))
    dt = time.perf_counter() - t0
    # the reference's own setting: Ceres num_threads = 4 (pose_graph.cpp:96)
    done4, t4 = 0, time.perf_counter()
    with ThreadPoolExecutor(4) as ex:
        while time.perf_counter() - t4 < max(2.0, seconds / 4):
            done4 += sum(ex.map(task, range(4)))
    dt4 = time.perf_counter() - t4
    out = {"value": done / dt / 1e6, "unit": "Mresiduals+Jacobians/s", "cores": cores,
           "value_4_threads": done4 / dt4 / 1e6,
           "kind": "port",
           "sample": f"constraint 0 of the same graph ({n} residuals, one 256^3 pair) evaluated "
                     f"{done // n} times, one evaluation per task on {cores} threads, {dt:.1f} s; "
                     "oracle/reg_oracle.c (" + orc.build_flags() + ")"}
    # The reference's OWN RegistrationCostFunction::Evaluate (oracle/_ref: its source compiled
    # against stand-in headers, hashed-block voxblox layer included), when the prebuilt library
    # travelled here: same constraint, same poses, its results checked against the port's.
    try:
        from oracle import ref_reg
        if ref_reg.available():
            subs_ref = {}
            for k in (a, b):
                bi, td, tw, ed, eo = subs[k]
                subs_ref[k] = ref_reg.Submap(k, true_poses[k], args.voxel_size, 16, bi, td, tw, ed, eo)
            # the reference walks its hash map in its own block order: same point SET, so compare sorted
            cf0 = ref_reg.RegistrationCostFunction(subs_ref[a], subs_ref[b])
            ok_r, r_ref, _, _ = cf0.Evaluate(poses[a], poses[b])
            ok_p, r_port, _, _ = orc.reg_evaluate(layer, xyz, dist, w, poses[a], poses[b])
            same = bool(ok_r and ok_p and np.array_equal(np.sort(r_ref), np.sort(r_port)))
            n_thr = min(cores, 64)
            cfs = [ref_reg.RegistrationCostFunction(subs_ref[a], subs_ref[b]) for _ in range(n_thr)]

            def ref_task(i):
                cfs[i].Evaluate(poses[a], poses[b])
                return cfs[i].num_residuals()

            def timed(threads, budget):
                cnt, t = 0, time.perf_counter()
                with ThreadPoolExecutor(threads) as ex:
                    while time.perf_counter() - t < budget:
                        cnt += sum(ex.map(ref_task, range(threads)))
                return cnt / (time.perf_counter() - t) / 1e6
            out["reference_source"] = {
                "kind": "reference", "unit": "Mresiduals+Jacobians/s",
                "value": timed(n_thr, max(2.0, seconds / 3)), "cores": n_thr,
                "value_4_threads": timed(4, max(2.0, seconds / 6)),
                "residuals_equal_to_port": same,
                "sample": "the same constraint through /root/reference's registration_cost_function.cpp, "
                          "compiled (g++ -O2) against oracle/ref_shims (hashed 16^3 blocks of 12/20-byte "
                          "voxels behind shared_ptr, minimal Eigen); one cost function per thread"}
    except Exception as e:                                    # the checker must never sink the bench
        out["reference_source"] = {"error": repr(e)}
    return out


def _room_points(dirs, origin):
    """first hit of unit rays from `origin` with the inside of a 10 x 8 x 4 m room"""
    lo, hi = np.array([-5.0, -4.0, -1.0]) - origin, np.array([5.0, 4.0, 3.0]) - origin
    with np.errstate(divide="ignore", invalid="ignore"):
        t = np.where(dirs > 0, hi / dirs, np.where(dirs < 0, lo / dirs, np.inf))
    return (dirs * t.min(1)[:, None]).astype(np.float32)


def tsdf_bench(capi, ctx, torch, scans=20, cpu_scans=2):
    """Second hot path (TSDF): whole scans resident in HBM, one integratePointCloud per
    scan into the active layer, HIP-event timed.  Two sensor shapes from BASELINE.json:
    RGB-D 640x480 @ 0.05 m voxels (config 4) and OS1-64-shaped LiDAR 64x1024 @ 0.20 m with
    the shipped yaml (config 2's integrator settings)."""
    from oracle import pyoracle as orc
    out = {}
    u, v = np.meshgrid((np.arange(640) - 319.5) / 525.0, (np.arange(480) - 239.5) / 525.0)
    d_rgbd = np.stack([np.ones_like(u), -u, -v], -1).reshape(-1, 3)
    d_rgbd /= np.linalg.norm(d_rgbd, axis=1, keepdims=True)
    az = np.linspace(-np.pi, np.pi, 1024, endpoint=False)
    el = np.deg2rad(np.linspace(-16.6, 16.6, 64))
    A, E = np.meshgrid(az, el)
    d_lidar = np.stack([np.cos(E) * np.cos(A), np.cos(E) * np.sin(A), np.sin(E)], -1).reshape(-1, 3)
    cases = {
        "rgbd_640x480_0.05m": (d_rgbd, 0.05, dict(default_truncation_distance=0.15, max_ray_length_m=5.0),
                               (-8, -6, -2), (16, 12, 7)),
        "lidar_64x1024_0.20m_voxgraph_yaml": (d_lidar, 0.20, dict(
            default_truncation_distance=0.60, max_ray_length_m=16.0, use_const_weight=1,
            use_weight_dropoff=1, use_sparsity_compensation_factor=1,
            sparsity_compensation_factor=20.0), (-3, -3, -2), (6, 6, 4)),
    }
    for name, (dirs, vs, kw, bmin, bdim) in cases.items():
        poses, clouds = [], []
        for k in range(scans):
            origin = np.array([-2.0 + 0.15 * k, 0.5 - 0.05 * k, 0.3 + 0.01 * k])
            yaw = 0.05 * k
            c, s_ = np.cos(yaw), np.sin(yaw)
            R = np.array([[c, -s_, 0], [s_, c, 0], [0, 0, 1.0]])
            pts_w = _room_points(dirs @ R.T, origin)            # hits, relative to the sensor, world axes
            pts_c = (pts_w @ R).astype(np.float32)              # sensor frame
            poses.append(np.array([np.cos(yaw / 2), 0, 0, np.sin(yaw / 2), *origin], np.float32))
            clouds.append(pts_c)
        n_pts = clouds[0].shape[0]
        reach = kw["max_ray_length_m"] + kw["default_truncation_distance"] + 2 * vs

        def new_layer():
            # unbounded layer; room for the whole sweep is reserved up front so that no timed
            # scan pays for an enlargement (scans would reserve for themselves otherwise)
            lay = capi.TsdfLayer(ctx, vs, 16)
            for k in (0, scans - 1):
                lay.reserve(poses[k][4:7], reach)
            return lay

        layer = new_layer()
        integ = capi.FastTsdfIntegrator(ctx, capi.tsdf_config(**kw), layer)
        dev = [torch.from_numpy(c_).cuda() for c_ in clouds]
        torch.cuda.synchronize()
        integ.integrate_device(poses[0], dev[0].data_ptr(), None, n_pts)        # warm-up scan
        ctx.synchronize()
        g0 = layer.growths()
        updates = 0
        ctx.timer_start()
        for k in range(1, scans):
            integ.integrate_device(poses[k], dev[k].data_ptr(), None, n_pts)
        ms = ctx.timer_stop()
        grew = layer.growths() - g0
        # second pass: voxel updates per scan (the count needs a sync per scan) and, with the stream
        # drained around every launch, the duration of each scan's kernel by itself (HIP events)
        layer2 = new_layer()
        integ.setLayer(layer2)
        kernel_ms = 0.0
        for k in range(scans):
            ctx.synchronize()
            ctx.timer_start()
            integ.integrate_device(poses[k], dev[k].data_ptr(), None, n_pts)
            t_k = ctx.timer_stop()
            kernel_ms += t_k if k >= 1 else 0.0
        kernel_ms /= (scans - 1)
        layer2b = new_layer()
        integ.setLayer(layer2b)
        for k in range(scans):
            u_ = integ.integrate_device(poses[k], dev[k].data_ptr(), None, n_pts, count=True)
            updates += u_ if k >= 1 else 0
        n_blocks, dropped = layer.stats()
        # heaviest case: the first scan into an empty layer with a fresh integrator (no
        # previously observed voxels: every ray runs to its early-out or to the sensor)
        layer3 = new_layer()
        integ3 = capi.FastTsdfIntegrator(ctx, capi.tsdf_config(**kw), layer3)
        ctx.synchronize()
        ctx.timer_start()
        integ3.integrate_device(poses[0], dev[0].data_ptr(), None, n_pts)
        first_ms = ctx.timer_stop()
        layer4 = new_layer()
        integ4 = capi.FastTsdfIntegrator(ctx, capi.tsdf_config(**kw), layer4)
        first_updates = integ4.integrate_device(poses[0], dev[0].data_ptr(), None, n_pts, count=True)
        for o in (integ3, integ4, layer3, layer4, layer2b):
            o.destroy()
        # voxblox's other integrator on the same scans: MergedTsdfIntegrator (one ray per end voxel)
        layer6 = new_layer()
        integ6 = capi.FastTsdfIntegrator(ctx, capi.tsdf_config(**kw), layer6)
        integ6.integrate_merged_device(poses[0], dev[0].data_ptr(), None, n_pts)
        ctx.synchronize()
        ctx.timer_start()
        for k in range(1, scans):
            integ6.integrate_merged_device(poses[k], dev[k].data_ptr(), None, n_pts)
        merged_ms = ctx.timer_stop() / (scans - 1)
        merged_updates = integ6.integrate_merged_device(poses[1], dev[1].data_ptr(), None, n_pts, count=True)
        merged_dropped = layer6.stats()[1]
        for o in (integ6, layer6):
            o.destroy()
        # the REPRODUCIBLE mode (vgx_tsdf_config.deterministic) on the same scans: wall clock per scan (the
        # mode synchronises with the host several times per scan), its voxel updates, and -- the TSDF
        # path's same-run parity evidence -- its layer after the CPU sample's scans against the oracle's
        layer7 = new_layer()
        integ7 = capi.FastTsdfIntegrator(ctx, capi.tsdf_config(deterministic=1, **kw), layer7)
        integ7.integrate_device(poses[0], dev[0].data_ptr(), None, n_pts)
        ctx.synchronize()
        d0 = time.perf_counter()
        for k in range(1, scans):
            integ7.integrate_device(poses[k], dev[k].data_ptr(), None, n_pts)
        ctx.synchronize()
        det_ms = (time.perf_counter() - d0) * 1e3 / (scans - 1)
        det_updates = integ7.integrate_device(poses[1], dev[1].data_ptr(), None, n_pts, count=True)
        for o in (integ7, layer7):
            o.destroy()
        # the drop-in call itself: host pointers (pageable), PCIe upload included, returns when done;
        # layer created the way voxblox creates one (no reservation at all)
        layer5 = capi.TsdfLayer(ctx, vs, 16)
        integ5 = capi.FastTsdfIntegrator(ctx, capi.tsdf_config(**kw), layer5)
        integ5.integratePointCloud(poses[0], clouds[0])
        h0 = time.perf_counter()
        for k in range(1, scans):
            integ5.integratePointCloud(poses[k], clouds[k])
        host_ms = (time.perf_counter() - h0) * 1e3 / (scans - 1)
        host_growths = layer5.growths()
        for o in (integ5, layer5):
            o.destroy()
        # CPU oracle on a bounded sample (single thread: the restatement is serial)
        ol = orc.TsdfLayer(vs, 16)
        oi = orc.FastTsdfIntegrator(orc.tsdf_config(**kw), ol)
        oi.integratePointCloud(poses[0], clouds[0])
        t0, cu = time.perf_counter(), 0
        for k in range(1, 1 + cpu_scans):
            cu += oi.integratePointCloud(poses[k], clouds[k])
        cdt = time.perf_counter() - t0
        # same scans through the reproducible mode: bit for bit the oracle's layer?
        layer8 = capi.TsdfLayer(ctx, vs, 16)
        integ8 = capi.FastTsdfIntegrator(ctx, capi.tsdf_config(deterministic=1, **kw), layer8)
        gu = 0
        for k in range(0, 1 + cpu_scans):
            u_ = integ8.integrate_device(poses[k], dev[k].data_ptr(), None, n_pts, count=True)
            gu += u_ if k >= 1 else 0
        obi, od, ow, oc = ol.download()
        gbi, gd, gw, gc = layer8.download()
        det_parity = {"scans": 1 + cpu_scans, "voxel_updates_equal": bool(gu == cu),
                      "blocks": int(len(obi)), "voxels_compared": int(od.size),
                      "bit_identical": bool(np.array_equal(obi, gbi) and np.array_equal(od.view(np.uint32), gd.view(np.uint32))
                                            and np.array_equal(ow.view(np.uint32), gw.view(np.uint32)) and np.array_equal(oc, gc)),
                      "checker": "oracle/tsdf_oracle.c (single thread, mixed order) [recalled: parity unpinned]"}
        for o in (integ8, layer8):
            o.destroy()
        # the same port on all host cores: the path does not shard (one active submap), so this is
        # REPLICAS -- one integrator + layer per thread, every thread the same scans
        from concurrent.futures import ThreadPoolExecutor
        cores = os.cpu_count() or 1
        reps = []
        for _ in range(cores):
            l_ = orc.TsdfLayer(vs, 16)
            i_ = orc.FastTsdfIntegrator(orc.tsdf_config(**kw), l_)
            reps.append((l_, i_))

        def replica(j):
            i_ = reps[j][1]
            i_.integratePointCloud(poses[0], clouds[0])
            t_ = time.perf_counter()
            for k in range(1, 1 + cpu_scans):
                i_.integratePointCloud(poses[k], clouds[k])
            return time.perf_counter() - t_
        t0r = time.perf_counter()
        with ThreadPoolExecutor(cores) as ex:
            rt = list(ex.map(replica, range(cores)))
        cpu_all = {"Mpoints_per_s": n_pts * cpu_scans * cores / max(rt) / 1e6, "cores": cores, "kind": "port",
                   "replicas": cores, "wall_s": time.perf_counter() - t0r,
                   "sample": f"{cores} replicas (one integrator + layer per thread) x {cpu_scans} scans; the "
                             "restatement is serial within a scan"}
        del reps
        timed = scans - 1
        alg_bytes_scan = 16.0 * n_pts + 24.0 * updates / timed
        out[name] = {"points_per_scan": n_pts, "scans_timed": timed, "ms_per_scan": ms / timed,
                     "Mpoints_per_s": n_pts * timed / ms / 1e3,
                     "Mvoxel_updates_per_s": updates / ms / 1e3,
                     "voxel_updates_per_scan": updates / timed, "blocks": n_blocks,
                     "dropped_updates": dropped, "layer_enlargements_in_timed_region": grew,
                     "algorithmic_GBs": alg_bytes_scan * timed / ms / 1e6,
                     # the TSDF kernel is latency bound (one dependent L2 round trip per DDA step of
                     # the longest ray), nowhere near the HBM roofline: reported for completeness
                     "roofline": {"bound": "hbm", "kernel": "tsdf_integrate_kernel<true>",
                                  "kernel_ms": kernel_ms,
                                  "kernel_ms_how": "HIP events around each scan's launch, stream drained before",
                                  "bytes_per_launch": alg_bytes_scan,
                                  "achieved": alg_bytes_scan / kernel_ms / 1e6, "peak": HBM_PEAK_GBS,
                                  "unit": "GB/s", "frac": alg_bytes_scan / kernel_ms / 1e6 / HBM_PEAK_GBS,
                                  "back_to_back_ms_per_scan": ms / timed,
                                  "back_to_back_over_kernel": (ms / timed) / kernel_ms},
                     "host_pointer_call": {"ms_per_scan": host_ms, "Mpoints_per_s": n_pts / host_ms / 1e3,
                                           "layer_enlargements": host_growths,
                                           "note": "vgx_tsdf_integrate into an unreserved layer: pageable host "
                                                   "points, PCIe upload, enlargements and completion wait included"},
                     "merged_integrator": {"ms_per_scan": merged_ms, "Mpoints_per_s": n_pts / merged_ms / 1e3,
                                           "roofline": {"bound": "hbm", "unit": "GB/s", "peak": HBM_PEAK_GBS,
                                                        "bytes_per_launch": 16.0 * n_pts + 24.0 * merged_updates,
                                                        "achieved": (16.0 * n_pts + 24.0 * merged_updates) / merged_ms / 1e6,
                                                        "frac": (16.0 * n_pts + 24.0 * merged_updates) / merged_ms / 1e6 / HBM_PEAK_GBS,
                                                        "time": "back-to-back scans, all kernels of a scan (keys, sort, "
                                                                "heads, rays)"},
                                           "voxel_updates_per_scan": merged_updates, "dropped_updates": merged_dropped,
                                           "Mvoxel_updates_per_s": merged_updates / merged_ms / 1e3,
                                           "note": "vgx_tsdf_integrate_merged_device: key + stable radix sort + group heads + "
                                                   "cooperative merge, then every ray written out, sorted by voxel and applied "
                                                   "voxel by voxel in group order (no early-out: the voxels next to the sensor "
                                                   "take one update per group, a sequential f32 chain)"},
                     "reproducible_mode": {"ms_per_scan": det_ms, "Mpoints_per_s": n_pts / det_ms / 1e3,
                                           "voxel_updates_per_scan": det_updates,
                                           "over_racing_kernel": det_ms / (ms / timed),
                                           "parity_vs_oracle": det_parity,
                                           "note": "vgx_tsdf_config.deterministic = 1: the single-thread visiting "
                                                   "order resolved in parallel (sort by approximate-set slot, "
                                                   "fixed-point sweeps, ordered per-voxel updates); wall clock incl. "
                                                   "its host synchronisations"},
                     "first_scan": {"ms": first_ms, "voxel_updates": first_updates,
                                    "Mvoxel_updates_per_s": first_updates / first_ms / 1e3,
                                    "algorithmic_GBs": (16.0 * n_pts + 24.0 * first_updates) / first_ms / 1e6},
                     "cpu_baseline": {"Mpoints_per_s": n_pts * cpu_scans / cdt / 1e6,
                                      "Mvoxel_updates_per_s": cu / cdt / 1e6, "cores": 1,
                                      "kind": "port", "sample": f"{cpu_scans} scans, oracle/tsdf_oracle.c ("
                                                                + orc.build_flags() + ")",
                                      "all_cores": cpu_all}}
        for o in (integ, layer, layer2):
            o.destroy()
    return out


def finish_bench(capi, ctx, args, true_poses):
    """finishSubmap() on the device for one 256^3 submap of the bench scene (HIP-event
    timed): ESDF from TSDF, kVoxels and kIsosurfacePoints extraction."""
    sm = capi.Submap.synth_city(ctx, 0, args.voxel_size, 16, args.block_min, args.block_dims,
                                args.truncation, args.esdf_max, 10.0, true_poses[0], args.seed)
    out = {}
    for name, fn in (("generate_esdf_ms", lambda: sm.generate_esdf()),
                     ("extract_voxel_points_ms", lambda: sm.extract_voxel_points(1.0, 0.3, True)),
                     ("extract_isosurface_points_ms", lambda: sm.extract_isosurface_points(1.0))):
        fn()
        ctx.synchronize()
        ctx.timer_start()
        r = fn()
        out[name] = ctx.timer_stop()
        out[name.replace("_ms", "_result")] = int(r)
    sm.destroy()
    return out


class GpuBackendLite:
    """single batch: fused pass + assembly + copy of the fused buffer to the host (what
    vgx_reg_multi_evaluate_fused returns), without the harness around it"""

    def __init__(self, capi, ctx, batch, n_nodes, torch):
        self.ctx, self.batch, self.n_nodes = ctx, batch, n_nodes
        self.buf = torch.zeros(capi.fused_size(n_nodes, batch.n_global), dtype=torch.float64, device="cuda")
        self.host = torch.zeros_like(self.buf, device="cpu").pin_memory()
        torch.cuda.current_stream().synchronize()
        self.torch = torch

    def __call__(self):
        self.batch.evaluate_normal(self._poses, to_host=False)
        self.batch.assemble(self.n_nodes, self.buf.data_ptr(), zero_first=True)
        self.ctx.synchronize()
        self.host.copy_(self.buf, non_blocking=True)
        self.torch.cuda.current_stream().synchronize()
        return self.host.numpy()


def config5_bench(capi, ctx, torch, dist, use_dist, rank, world, args):
    """BASELINE configs[4]: 1000 submaps @ 128^3 on a loop (serpentine) trajectory, odometry edges
    with accumulated drift, 20 injected loop-closure relative-pose edges
    (PoseGraphInterface::addLoopClosureMeasurement, pose_graph_interface.cpp:68-92) and the
    reference's two-stage optimisation (PoseGraphInterface::optimize, :177-198: loop closures are
    new, so first optimise WITHOUT the registration constraints, then with all of them).
    Constraints pair-sharded over the ranks; one all-reduce per solver evaluation."""
    from harness import lm
    from harness.backends import GpuBackend
    n_lanes, per_lane = args.config5_grid
    n = n_lanes * per_lane
    rng = np.random.default_rng(4)                                 # seed 4
    vs, dims, bmin = 0.2, (8, 8, 8), (-4, -4, -2)                   # 128^3 voxels, 25.6 m cubes
    dx, dy = 12.8, 19.2                                             # 50 % overlap along a lane, 25 % across

    def idx(lane, q):                                               # path index of x-position q in a lane
        return lane * per_lane + (q if lane % 2 == 0 else per_lane - 1 - q)
    true = np.zeros((n, 4))
    for lane in range(n_lanes):
        for q in range(per_lane):
            true[idx(lane, q)] = [q * dx, lane * dy, 0.0, rng.uniform(-0.1, 0.1)]
    pairs = [(i, i + 1) for i in range(n - 1)]
    for lane in range(n_lanes - 1):
        for q in range(per_lane):
            for dq in (-1, 0, 1):
                if 0 <= q + dq < per_lane:
                    a, b = idx(lane, q), idx(lane + 1, q + dq)
                    if abs(a - b) > 1:
                        pairs.append((min(a, b), max(a, b)))
    pairs = np.array(sorted(set(pairs)), np.int32)

    def between(pa, pb):
        c, s_ = np.cos(pa[3]), np.sin(pa[3])
        d = pb[:3] - pa[:3]
        return np.array([c * d[0] + s_ * d[1], -s_ * d[0] + c * d[1], d[2], lm.normalize_angle(pb[3] - pa[3])])

    def compose(pose, delta):
        c, s_ = np.cos(pose[3]), np.sin(pose[3])
        return np.array([pose[0] + c * delta[0] - s_ * delta[1], pose[1] + s_ * delta[0] + c * delta[1],
                         pose[2] + delta[2], lm.normalize_angle(pose[3] + delta[3])])
    # odometry: good in z and yaw (the yaml's information 2500), drifting in x, y.  The per-step noise
    # is sized so that neighbours across lanes (40-80 steps apart along the path) start within the
    # registration basin (a few voxels), as they do when voxgraph optimises after every new submap
    sig = np.array([0.01, 0.01, 0.001, 5e-5])
    info_odo = [1.0, 1.0, 2500.0, 2500.0]                            # voxgraph_mapper.yaml:41-47
    info_lc = [100.0, 100.0, 2500.0, 2500.0]                         # not in the yaml (template is zero): 0.1 m
    poses0 = true[:1].copy()
    edges = []
    for k in range(n - 1):
        delta = between(true[k], true[k + 1]) + rng.normal(0, sig)
        edges.append(lm.RelativePoseEdge(k, k + 1, delta[:3], delta[3], info_odo))
        poses0 = np.vstack([poses0, compose(poses0[k], delta)])
    n_lc = 20
    for j in range(n_lc):
        lane = 1 + (j * (n_lanes - 1)) // n_lc
        q = (7 * j + 3) % per_lane
        a, b = idx(lane - 1, q), idx(lane, q)
        # a loop closure's yaw error acts over the whole lane behind it (1 mrad over 500 m = 0.5 m), so
        # a usable one is accurate to a fraction of that
        delta = between(true[a], true[b]) + rng.normal(0, [0.03, 0.03, 0.005, 1e-4])
        edges.append(lm.RelativePoseEdge(a, b, delta[:3], delta[3], info_lc))

    t0 = time.perf_counter()
    submaps, n_points = [], []
    for k in range(n):
        sm = capi.Submap.synth_city(ctx, k, vs, 16, bmin, dims, args.truncation, args.esdf_max, 10.0,
                                    true[k], args.seed)
        n_points.append(sm.extract_voxel_points(1.0, 0.3, True))
        sm.release_raw_layers()
        submaps.append(sm)
    ctx.synchronize()
    setup_s = time.perf_counter() - t0
    cfg = capi.default_config(registration_point_type=capi.POINTS_VOXELS)
    mine = lpt_shards([n_points[a] for a, _ in pairs], world)[rank]
    cfs = [capi.RegistrationCostFunction(ctx, submaps[pairs[c][0]], submaps[pairs[c][1]], cfg) for c in mine]
    batch = capi.RegistrationBatch(ctx, cfs, pairs[mine], global_index=mine, n_global=len(pairs))
    backend = GpuBackend(capi, ctx, batch, n, dist if use_dist else None)

    def barrier():
        if use_dist:
            dist.barrier()

    def rmse(p):
        return float(np.sqrt(((p[:, :3] - true[:, :3]) ** 2).sum(1).mean()))

    def rmse_aligned(p):
        """absolute trajectory error after the best rigid alignment (yaw + translation) of the whole
        estimate onto the truth: what is left once the gauge -- which submap 0 alone holds, through the
        few constraints it takes part in -- is taken out"""
        a, b = p[:, :2] - p[:, :2].mean(0), true[:, :2] - true[:, :2].mean(0)
        H = a.T @ b
        th = np.arctan2(H[0, 1] - H[1, 0], H[0, 0] + H[1, 1])
        R = np.array([[np.cos(th), -np.sin(th)], [np.sin(th), np.cos(th)]])
        dxy = a @ R.T - b
        dz = (p[:, 2] - p[:, 2].mean()) - (true[:, 2] - true[:, 2].mean())
        return float(np.sqrt((dxy ** 2).sum(1).mean() + (dz ** 2).mean()))
    kw = dict(parameter_tolerance=1e-10, max_seconds=1e9)            # Ceres-default function_tolerance decides
    if os.environ.get("VGX_C5_DEBUG"):
        def parts(p):
            reg = float(backend(p)[0]) * 0.5
            tot = lm.Problem(backend, n, pairs, edges).evaluate_reduced(p)[0]
            return {"registration": reg, "edges": tot - reg, "rmse": rmse(p)}
        print("C5 at truth", parts(true), file=sys.stderr)
        print("C5 at odometry", parts(poses0), file=sys.stderr)
        xt, st = lm.solve(lm.Problem(backend, n, pairs, edges), true, **kw)
        print("C5 from truth ->", parts(xt), st["iterations"], st["termination"], file=sys.stderr)
        xo, so = lm.solve(lm.Problem(backend, n, pairs, edges), poses0, **kw)
        print("C5 from odometry (no stage 1) ->", parts(xo), so["iterations"], so["termination"], file=sys.stderr)
        err = np.linalg.norm((xo - true)[:, :2], axis=1)
        print("C5 error by lane", [round(float(err[l * per_lane:(l + 1) * per_lane].mean()), 3) for l in range(n_lanes)], file=sys.stderr)
        print("C5 aligned rmse: odometry", rmse_aligned(poses0), "from odometry ->", rmse_aligned(xo), file=sys.stderr)
        reg_only = lm.Problem(backend, n, pairs, [])
        xr, sr = lm.solve(reg_only, poses0, **kw)
        print("C5 registration only from odometry ->", rmse(xr), sr["iterations"], sr["termination"], file=sys.stderr)
        import contextlib
        with contextlib.redirect_stdout(sys.stderr):                 # the two-stage solve, step by step
            lm.optimize_two_stage(backend, n, pairs, edges, poses0, True, verbose=True, **kw)
    lm.optimize_two_stage(backend, n, pairs, edges, poses0, True, **kw)      # untimed warm-up
    torch.cuda.synchronize()
    barrier()
    s0 = time.perf_counter()
    x, summaries = lm.optimize_two_stage(backend, n, pairs, edges, poses0, True, **kw)
    torch.cuda.synchronize()
    barrier()
    sdt = torch.tensor([time.perf_counter() - s0], dtype=torch.float64, device="cuda")
    # stage 1 alone, for the intermediate error
    x1, _ = lm.solve(lm.Problem(lm.zero_registration_backend(n, len(pairs)), n, pairs, edges), poses0, **kw)
    # one fused evaluation of every registration constraint at the initial guess, timed by itself
    for _ in range(2):
        backend(poses0)
    torch.cuda.synchronize()
    barrier()
    e0 = time.perf_counter()
    for _ in range(10):
        backend(poses0)
    torch.cuda.synchronize()
    barrier()
    edt = torch.tensor([(time.perf_counter() - e0) / 10], dtype=torch.float64, device="cuda")
    rs = torch.tensor([float(batch.num_residuals())], dtype=torch.float64, device="cuda")
    if use_dist:
        dist.all_reduce(sdt, op=dist.ReduceOp.MAX)
        dist.all_reduce(edt, op=dist.ReduceOp.MAX)
        dist.all_reduce(rs, op=dist.ReduceOp.SUM)
    out = {"workload": f"configs[4]: {n} submaps @ 128^3 (0.2 m) on a serpentine loop trajectory "
                       f"({n_lanes} lanes x {per_lane}), {len(pairs)} registration constraints (kVoxels, all points), "
                       f"{n - 1} odometry edges with accumulated drift, {n_lc} injected loop-closure edges; "
                       "two-stage optimisation (pose_graph_interface.cpp:177-198)",
           "submaps": n, "registration_constraints": int(len(pairs)), "loop_closures": n_lc,
           "residuals_per_evaluation": float(rs.item()),
           "solve_ms": float(sdt.item()) * 1e3,
           "solve_gpu_evaluation_ms": sum(s_["backend_seconds"] for s_ in summaries) * 1e3,
           "solve_host_linear_algebra_ms": sum(s_["host_linear_algebra_seconds"] for s_ in summaries) * 1e3,
           "stage1_without_registration": {k: summaries[0][k] for k in ("iterations", "evaluations", "termination")},
           "stage2_all_constraints": {k: summaries[1][k] for k in ("iterations", "evaluations", "termination",
                                                                  "initial_cost", "final_cost")},
           # Past the first few iterations stage 2 walks at random among the kinks of the trilinear field
           # (steps of 1e-5 m, gain ratios between -50 and +100: VGX_C5_DEBUG=1 prints them), so WHEN
           # function_tolerance or the iteration cap ends it depends on the last bits of the sums: the
           # iteration count, and with it solve_ms, is not a property of the kernels.  These two are:
           "stage2_ms_per_iteration": summaries[1]["seconds"] * 1e3 / max(summaries[1]["iterations"], 1),
           "stage2_iterations_to_within_1e-3_of_final_cost": next(
               (int(it_) for it_, c_ in summaries[1]["cost_history"] if c_ <= summaries[1]["final_cost"] * (1 + 1e-3)), None),
           "registration_evaluation_ms": float(edt.item()) * 1e3,
           "position_rmse_m_odometry": rmse(poses0), "position_rmse_m_after_stage1": rmse(x1),
           "position_rmse_m_after": rmse(x),
           "position_rmse_m_aligned_odometry": rmse_aligned(poses0),
           "position_rmse_m_aligned_after_stage1": rmse_aligned(x1),
           "position_rmse_m_aligned_after": rmse_aligned(x),
           "rmse_note": "position_rmse_m_*: in the frame of the fixed first submap (the reference's gauge, "
                        "pose_graph_interface.cpp:30-32); *_aligned_*: after the best rigid alignment of the "
                        "whole estimate onto the truth (the registration cost is invariant to that transform "
                        "except through submap 0's own few constraints)",
           "stop_rule": "function_tolerance 1e-6 (Ceres default) in both stages, parameter_tolerance off",
           "parallelism": f"pair-sharded x{world} (LPT), submaps replicated, one all-reduce of "
                          f"{capi.fused_size(n, len(pairs)) * 8} B per evaluation",
           "setup_s": setup_s,
           "solver": "harness/lm.py (LM, banded Cholesky on the host; Ceres absent)"}
    if rank == 0 and not args.no_parity:
        from harness import parity_gate
        t_par = time.perf_counter()

        def layers_of(k):
            sm = capi.Submap.synth_city(ctx, k, vs, 16, bmin, dims, args.truncation, args.esdf_max, 10.0, true[k], args.seed)
            td, tw, ed, eo = sm.download_layers(16)
            bi = sm.block_index()
            sm.destroy()
            return bi, td, tw, ed, eo
        live_each = batch.count_live_each(poses0)
        chosen = parity_gate.choose(np.diff(batch.row_offsets()), live_each, n_total=8, n_partial=3, n_dead=1)
        e = parity_gate.check(capi, ctx, torch, "config 5 (the timed batch at the initial poses)", layers_of, submaps,
                              batch, pairs[mine], poses0, chosen, None, vs)
        e["seconds"] = time.perf_counter() - t_par
        out["parity"] = e
    for o in [batch] + cfs + submaps:
        o.destroy()
    return out


def multi_context_bench(capi, ctx0, torch, args, devices, submaps0, true_poses, pairs, weights, poses, cfg, single_batch,
                        n_sub, n_con):
    """vgx_reg_multi_evaluate_fused over len(devices) contexts (context 0 = ctx0, whose submaps are
    resident already; every other context gets the submaps its LPT share of the constraints needs)."""
    n_ctx = len(devices)
    t_setup = time.perf_counter()
    ctxs = [ctx0] + [capi.Context(d) for d in devices[1:]]
    shard_of = capi.lpt_shards(weights, n_ctx)
    subs = [dict(enumerate(submaps0))] + [dict() for _ in range(n_ctx - 1)]
    for k_ctx in range(1, n_ctx):
        need = sorted({int(s_) for c in range(n_con) if shard_of[c] == k_ctx for s_ in pairs[c]})
        for k in need:
            sm = capi.Submap.synth_city(ctxs[k_ctx], k, args.voxel_size, 16, args.block_min, args.block_dims,
                                        args.truncation, args.esdf_max, 10.0, true_poses[k], args.seed)
            sm.extract_voxel_points(1.0, 0.3, True)
            sm.release_raw_layers()
            subs[k_ctx][k] = sm
    cfs_m = [capi.RegistrationCostFunction(ctxs[shard_of[c]], subs[shard_of[c]][int(a)], subs[shard_of[c]][int(b)], cfg)
             for c, (a, b) in enumerate(pairs)]
    multi = capi.RegistrationMulti(ctxs, cfs_m, pairs)
    setup_s = time.perf_counter() - t_setup
    for _ in range(2):
        fused_m, _ = multi.evaluate_fused(poses)
    m0 = time.perf_counter()
    for _ in range(args.steps):
        fused_m, _ = multi.evaluate_fused(poses)
    m_ms = (time.perf_counter() - m0) / args.steps * 1e3
    out = {"contexts": n_ctx, "devices": len(set(devices)), "device_ids": devices,
           "what": "vgx_reg_multi_evaluate_fused (LPT shard by bytes moved, one host thread per context, event-ordered "
                   "fixed-order sum on context 0 over peer mappings, result on the host)",
           "ms_per_evaluation": m_ms,
           "Mresiduals_per_s": float(sum(cf.num_residuals() for cf in cfs_m)) / m_ms / 1e3,
           "constraints_per_context": [int((shard_of == k).sum()) for k in range(n_ctx)],
           "cost": float(fused_m[0]), "setup_s": setup_s}
    if len(set(devices)) == n_ctx and n_ctx > 1:
        # SURVEY.md 8(e) "compare": the same evaluation with ONE ncclAllReduce of the fused buffer instead of
        # the fixed-order sum over peer mappings
        try:
            multi.set_reduction(True)
            for _ in range(2):
                fused_r, _ = multi.evaluate_fused(poses)
            r0 = time.perf_counter()
            for _ in range(args.steps):
                fused_r, _ = multi.evaluate_fused(poses)
            out["rccl_allreduce"] = {"ms_per_evaluation": (time.perf_counter() - r0) / args.steps * 1e3,
                                     "max_rel_diff_vs_peer_sum": float(np.abs(fused_r - fused_m).max() / np.abs(fused_m).max()),
                                     "what": "vgx_reg_multi_set_reduction(VGX_REDUCE_RCCL): ncclAllReduce(sum, f64) in place "
                                             "on every context's stream, one communicator per context"}
            multi.set_reduction(False)
        except Exception as e:                                   # never sink the line on the optional variant
            out["rccl_allreduce"] = {"error": repr(e)}
    if single_batch is not None:
        single = GpuBackendLite(capi, ctx0, single_batch, n_sub, torch)
        single._poses = poses
        for _ in range(2):
            ref_buf = single()
        s0_ = time.perf_counter()
        for _ in range(args.steps):
            ref_buf = single()
        out["single_batch_ms_per_evaluation"] = (time.perf_counter() - s0_) / args.steps * 1e3
        out["single_batch_what"] = "the single batch (evaluate + assemble + copy to the host) on context 0 alone"
        out["max_rel_diff_vs_single_batch"] = float(np.abs(fused_m - ref_buf).max() / np.abs(ref_buf).max())
    multi.destroy()
    for o in cfs_m:
        o.destroy()
    for k_ctx in range(1, n_ctx):
        for sm in subs[k_ctx].values():
            sm.destroy()
        ctxs[k_ctx].close()
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--inner", type=int, default=25,
                    help="passes over all constraints per step (the driver's 20 steps then time > 2 s)")
    ap.add_argument("--no-parity", action="store_true",
                    help="skip the same-run parity gate (harness/parity_gate.py: a sample of the timed launches' "
                         "constraints against the reference source / the oracle)")
    ap.add_argument("--overlap-copies", type=int, default=6,
                    help="full-overlap workload: perturbed duplicates per submap (config 3 has ~6 "
                         "constraints per reference submap)")
    ap.add_argument("--no-full-overlap", action="store_true")
    ap.add_argument("--no-fo-plain", action="store_true",
                    help="skip the plain-launch-order measurement of the full-overlap workload (PMC passes: "
                         "one launch order per grid size)")
    ap.add_argument("--no-shipped", action="store_true",
                    help="skip the shipped-yaml evaluation (isosurface points, sampling_ratio 0.05, mirrored)")
    ap.add_argument("--grid", type=int, nargs=2, default=[20, 10], help="submap grid (200 submaps)")
    ap.add_argument("--block-dims", type=int, nargs=3, default=[16, 16, 16], help="256^3 voxels")
    ap.add_argument("--block-min", type=int, nargs=3, default=[-8, -8, -4])
    ap.add_argument("--voxel-size", type=float, default=0.2)
    ap.add_argument("--truncation", type=float, default=0.6)
    ap.add_argument("--esdf-max", type=float, default=2.0)
    ap.add_argument("--pose-sigma", type=float, default=0.3)
    ap.add_argument("--yaw-sigma", type=float, default=0.05)
    ap.add_argument("--seed", type=int, default=2)
    ap.add_argument("--cpu-seconds", type=float, default=12.0)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-fused", action="store_true")
    ap.add_argument("--no-solve", action="store_true")
    ap.add_argument("--no-tsdf", action="store_true")
    ap.add_argument("--pipeline", action="store_true",
                    help="run the config-2 stand-in at full length (30 submaps x 100 scans, harness/pipeline.py) "
                         "instead of the bounded one (10 submaps x 30 scans) the default line carries")
    ap.add_argument("--no-config2", action="store_true")
    ap.add_argument("--no-multi-ctx", action="store_true")
    ap.add_argument("--no-quad", action="store_true",
                    help="skip the shipped configuration's second measurement on quad bricks")
    ap.add_argument("--inprocess", action="store_true",
                    help="ONE process drives --gpus N devices through vgx_reg_multi_* (the product's in-process "
                         "multi-GPU component); launch WITHOUT torch.distributed.run.  The headline loop then "
                         "runs on device 0 alone and `multi_context` carries the N-GPU evaluation")
    ap.add_argument("--no-config5", action="store_true")
    ap.add_argument("--config", type=int, default=3, choices=[3, 5],
                    help="5: only BASELINE configs[4] (1000 submaps @ 128^3, loop closures, two-stage solve)")
    ap.add_argument("--config5-grid", type=int, nargs=2, default=[25, 40], help="lanes x submaps per lane")
    ap.add_argument("--calibrate", action="store_true",
                    help="PMC calibration: first launch evaluates poses 10 km apart, so every "
                         "evaluation reads exactly 20 B and writes exactly 36 B (profiles/README.md)")
    args = ap.parse_args()

    lib_path = os.path.join(ROOT, "voxgraph_amd", "lib", "libvoxgraph_amd.so")
    if not os.path.exists(lib_path):            # checkout without the (git-ignored) built library
        if int(os.environ.get("LOCAL_RANK", "0")) == 0:
            import __graft_entry__
            __graft_entry__.build()
        else:
            t_wait = time.time()
            while not os.path.exists(lib_path) and time.time() - t_wait < 900:
                time.sleep(1.0)
            time.sleep(2.0)                      # let the linker finish writing
    from voxgraph_amd import capi
    capi.load()
    import torch
    import torch.distributed as dist

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus and not (args.inprocess and world == 1):
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run "
                         "(or pass --inprocess for the one-process multi-GPU component)")
    # VGX_BENCH_DRYRUN=gloo: every rank on cuda:0 with the gloo backend, to walk the N>1
    # code path on a one-GPU box (profiles/README.md); never used for reported numbers
    dryrun = os.environ.get("VGX_BENCH_DRYRUN", "")
    if dryrun:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    # under torch.distributed.run (RANK set) the RCCL group is always created, also for
    # one rank, so the collective code path below is the one that runs at every N
    use_dist = world > 1 or "RANK" in os.environ
    if use_dist:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        os.environ.setdefault("RANK", "0")
        os.environ.setdefault("WORLD_SIZE", "1")
        if dryrun:
            dist.init_process_group(dryrun)
        else:
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))

    def barrier():
        if use_dist:
            dist.barrier()

    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        from oracle import pyoracle
        pyoracle.use_native_build()      # the CPU port as BASELINE.md times it: -O3 -march=native, built on this host
    ctx = capi.Context(local_rank)
    # one explicit (non-null) stream shared by the HIP library, torch ops and RCCL,
    # so kernel -> all-reduce ordering is by stream order, not by host syncs
    stream = torch.cuda.Stream()
    torch.cuda.set_stream(stream)
    ctx.set_stream(stream.cuda_stream)

    if args.config == 5:
        c5 = config5_bench(capi, ctx, torch, dist, use_dist, rank, world, args)
        if rank == 0:
            print(json.dumps({"metric": "full pose-graph solve ms (1000 submaps, two-stage)", "value": c5["solve_ms"],
                              "unit": "ms", "n_gpus": world, "higher_is_better": False, "scaling": "strong",
                              "vs_baseline": None, "dtype": "f32", "data": "synthetic", "config": c5}))
        if use_dist:
            dist.barrier()
            dist.destroy_process_group()
        return

    true_poses, poses, pairs = build_graph(args)
    n_sub, n_con = len(true_poses), len(pairs)
    global PROFILE_TRAFFIC
    tpath = os.path.join(ROOT, "profiles", "hbm_traffic.json")
    PROFILE_TRAFFIC = json.load(open(tpath)) if os.path.exists(tpath) else {}

    # ---- resident inputs: every rank holds every finished submap -------------
    t_setup = time.perf_counter()
    submaps, n_points, n_iso = [], [], []
    for k in range(n_sub):
        sm = capi.Submap.synth_city(ctx, k, args.voxel_size, 16, args.block_min, args.block_dims,
                                    args.truncation, args.esdf_max, 10.0, true_poses[k], args.seed)
        n_points.append(sm.extract_voxel_points(1.0, 0.3, True))      # voxgraph_submap.h:27-28
        if not args.no_shipped:
            n_iso.append(sm.extract_isosurface_points(1.0))            # finishSubmap(), voxgraph_submap.cpp:97
        sm.release_raw_layers()
        submaps.append(sm)
    ctx.synchronize()
    setup_s = time.perf_counter() - t_setup

    cfg = capi.default_config(registration_point_type=capi.POINTS_VOXELS)
    weights = np.array([n_points[a] for a, _ in pairs], np.int64)
    if world > 1 or not args.no_multi_ctx:
        # what a constraint costs is the bytes it moves at these poses, not its residual count: 36 B per
        # row written + ~45 B per residual in a chunk that can touch the reading submap
        # (vgx_reg_batch_count_live_each; profiles/shard_balance.py: slowest-shard balance at N = 8
        # 0.980 -> 0.991).  Every rank computes the same weights from the same replicated inputs.
        cfs_all = [capi.RegistrationCostFunction(ctx, submaps[a], submaps[b], cfg) for a, b in pairs]
        probe = capi.RegistrationBatch(ctx, cfs_all, pairs)
        weights = 36 * weights + 45 * probe.count_live_each(poses)
        probe.destroy()
        for cf in cfs_all:
            cf.destroy()
    weights_bytes = weights
    mine = lpt_shards(weights, world)[rank]
    cfs = [capi.RegistrationCostFunction(ctx, submaps[pairs[c][0]], submaps[pairs[c][1]], cfg)
           for c in mine]
    batch = capi.RegistrationBatch(ctx, cfs, pairs[mine], global_index=mine, n_global=n_con)
    R = batch.num_residuals()

    # ---- second workload: ~100 % correspondence (gather-dominated regime) ---------
    # submap k against K perturbed duplicates of itself; node n_sub * (1 + p) + k carries the
    # p-th perturbed pose of submap k.  Order (p, k): consecutive constraints share nothing.
    fo = None
    if not args.no_full_overlap:
        K = args.overlap_copies
        rng_fo = np.random.default_rng(args.seed + 100)
        poses_fo = np.concatenate([true_poses] + [
            true_poses + np.concatenate([rng_fo.normal(0, args.pose_sigma, (n_sub, 3)),
                                         rng_fo.normal(0, args.yaw_sigma, (n_sub, 1))], axis=1)
            for _ in range(K)])
        pairs_fo = np.array([(k, n_sub * (1 + p) + k) for p in range(K) for k in range(n_sub)], np.int32)
        mine_fo = lpt_shards([n_points[a] for a, _ in pairs_fo], world)[rank]
        cfs_fo = [capi.RegistrationCostFunction(ctx, submaps[pairs_fo[c][0]], submaps[pairs_fo[c][0]], cfg)
                  for c in mine_fo]
        batch_fo = capi.RegistrationBatch(ctx, cfs_fo, pairs_fo[mine_fo], global_index=mine_fo,
                                          n_global=len(pairs_fo))
        fo = {"batch": batch_fo, "poses": poses_fo, "R": batch_fo.num_residuals(), "n": len(pairs_fo)}

    R_buf = max(R, fo["R"] if fo else 0)
    residuals = torch.empty(R_buf, dtype=torch.float32, device="cuda")
    jac_ref = torch.empty((R_buf, 4), dtype=torch.float32, device="cuda")
    jac_read = torch.empty((R_buf, 4), dtype=torch.float32, device="cuda")

    def one_pass():
        batch.evaluate_points(poses, residuals.data_ptr(), jac_ref.data_ptr(), jac_read.data_ptr())

    def step():
        for _ in range(args.inner):
            one_pass()

    if args.calibrate:
        far = poses.copy()
        far[:, 0] += 1e4 * np.arange(n_sub)
        # twice: the first dispatch of a process has been seen to under-count FETCH_SIZE
        # (profiles/README.md); summarize.py calibrates on the second
        for _ in range(2):
            batch.evaluate_points(far, residuals.data_ptr(), jac_ref.data_ptr(), jac_read.data_ptr())
            torch.cuda.synchronize()
        assert int((jac_ref[:R].abs().sum(dim=1) > 0).sum().item()) == 0
    for _ in range(args.warmup):
        step()
    torch.cuda.synchronize()
    barrier()
    ctx.timer_start()                      # HIP events on the stream the kernel runs on
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    kernel_ms = ctx.timer_stop() / max(args.steps * args.inner, 1)
    torch.cuda.synchronize()
    barrier()
    dt = time.perf_counter() - t0
    tmax = torch.tensor([dt], dtype=torch.float64, device="cuda")
    rtot = torch.tensor([float(R)], dtype=torch.float64, device="cuda")
    kmax = torch.tensor([kernel_ms], dtype=torch.float64, device="cuda")
    if use_dist:
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dist.all_reduce(rtot, op=dist.ReduceOp.SUM)
        dist.all_reduce(kmax, op=dist.ReduceOp.MAX)
    dt, total_evals, kernel_ms_max = float(tmax.item()), float(rtot.item()), float(kmax.item())

    # correspondences on this rank (for the conservative byte count)
    with_corr = int((jac_ref[:R].abs().sum(dim=1) > 0).sum().item())
    checksum = float(residuals[:R].double().pow(2).sum().item())
    wc_tot = torch.tensor([float(with_corr)], dtype=torch.float64, device="cuda")
    if use_dist:
        dist.all_reduce(wc_tot, op=dist.ReduceOp.SUM)
    with_corr_total = float(wc_tot.item())

    # ---- parity gate, config 3: rows of the LAST TIMED PASS (still in the output buffers) ----
    parity_entries = []

    def layers_of_config3(k):
        sm = capi.Submap.synth_city(ctx, k, args.voxel_size, 16, args.block_min, args.block_dims,
                                    args.truncation, args.esdf_max, 10.0, true_poses[k], args.seed)
        td, tw, ed, eo = sm.download_layers(16)
        bi = sm.block_index()
        sm.destroy()
        return bi, td, tw, ed, eo
    if rank == 0 and not args.no_parity:
        from harness import parity_gate
        t_par = time.perf_counter()
        live_each = batch.count_live_each(poses)
        chosen = parity_gate.choose(np.diff(batch.row_offsets()), live_each, n_total=8, n_partial=3, n_dead=1)
        e = parity_gate.check(capi, ctx, torch, "config 3 (rows of the last timed pass)", layers_of_config3, submaps, batch,
                              pairs[mine], poses, chosen, (residuals, jac_ref, jac_read), args.voxel_size)
        e["partly_culled_constraints"] = int(sum(1 for c in chosen if 0 < live_each[c] < 0.9 * np.diff(batch.row_offsets())[c]))
        e["seconds"] = time.perf_counter() - t_par
        parity_entries.append(e)

    # points the materialising kernel reads at these poses: tiles whose chunks all miss the reading
    # submap's block box write zeros without reading (vgx_reg.hip reg_eval_points_body); where the
    # launch order groups the constraints of a reference submap on one XCD they read its points once
    def points_read(bt, ps):
        lv, uq = bt.count_live(ps, unique=True)
        grouped = bt.launch_order(points_pass=True)
        return {"live": int(lv), "distinct": int(uq), "grouped": int(grouped),
                "read": int(uq if grouped == 1 else lv)}
    pr = points_read(batch, poses)

    # ---- full-overlap workload, timed the same way (HIP events on the kernel's stream) ----
    fo_out = None
    if fo:
        def time_fo(bt):
            def fo_pass():
                bt.evaluate_points(fo["poses"], residuals.data_ptr(), jac_ref.data_ptr(), jac_read.data_ptr())
            n_fo = max(args.steps, 1)
            for _ in range(2):
                fo_pass()
            torch.cuda.synchronize()
            barrier()
            ctx.timer_start()
            f0 = time.perf_counter()
            for _ in range(n_fo):
                fo_pass()
            k_ms = ctx.timer_stop() / n_fo
            torch.cuda.synchronize()
            barrier()
            return n_fo, time.perf_counter() - f0, k_ms
        n_fo, fo_dt, fo_kernel_ms = time_fo(fo["batch"])
        fo_corr = int((jac_ref[:fo["R"]].abs().sum(dim=1) > 0).sum().item())
        fo_pr = points_read(fo["batch"], fo["poses"])
        # the same constraints in plain constraint-major launch order (nothing shared through the L2:
        # the regime the 88 B-per-evaluation contract figure describes); the order is chosen at a
        # batch's first evaluation, VGX_POINTS_TILE_ORDER is read then
        fo_plain_ms = None
        if not args.no_fo_plain:
            prev = os.environ.get("VGX_POINTS_TILE_ORDER")
            os.environ["VGX_POINTS_TILE_ORDER"] = "0"
            plain = capi.RegistrationBatch(ctx, cfs_fo, pairs_fo[mine_fo], global_index=mine_fo, n_global=len(pairs_fo))
            _, _, fo_plain_ms = time_fo(plain)
            assert plain.launch_order(points_pass=True) == 0
            plain.destroy()
            if prev is None:
                del os.environ["VGX_POINTS_TILE_ORDER"]
            else:
                os.environ["VGX_POINTS_TILE_ORDER"] = prev
        red = torch.tensor([fo_dt, fo_kernel_ms, fo_plain_ms or 0.0], dtype=torch.float64, device="cuda")
        tot = torch.tensor([float(fo["R"]), float(fo_corr)], dtype=torch.float64, device="cuda")
        if use_dist:
            dist.all_reduce(red, op=dist.ReduceOp.MAX)
            dist.all_reduce(tot, op=dist.ReduceOp.SUM)
        fo_out = {"dt": float(red[0].item()), "kernel_ms_max": float(red[1].item()), "kernel_ms": fo_kernel_ms,
                  "plain_kernel_ms": fo_plain_ms, "plain_kernel_ms_max": float(red[2].item()) if fo_plain_ms else None,
                  "passes": n_fo, "R": fo["R"], "R_total": float(tot[0].item()), "with_corr": fo_corr,
                  "with_corr_total": float(tot[1].item()), "points": fo_pr}

    if fo and rank == 0 and not args.no_parity:
        t_par = time.perf_counter()
        fo["batch"].evaluate_points(fo["poses"], residuals.data_ptr(), jac_ref.data_ptr(), jac_read.data_ptr())
        torch.cuda.synchronize()
        n_fo_local = len(mine_fo)
        chosen = [int(c) for c in np.unique(np.linspace(0, n_fo_local - 1, 4).round().astype(int))]
        e = parity_gate.check(capi, ctx, torch, "full overlap (one more launch of the timed batch)", layers_of_config3,
                              submaps, fo["batch"], pairs_fo[mine_fo], fo["poses"], chosen,
                              (residuals, jac_ref, jac_read), args.voxel_size, submap_of_node=lambda nd: nd % n_sub)
        e["seconds"] = time.perf_counter() - t_par
        parity_entries.append(e)

    # ---- fused pass + all-reduce (solver-iteration form), reported separately --
    def fused_bench(bt, ps, n_nodes, n_global, corr_total, evals_total, ref_cost, traffic_key):
        size = capi.fused_size(n_nodes, n_global)
        tr = (PROFILE_TRAFFIC.get("fused") or {}).get(traffic_key) or {}
        tr_bytes = tr.get("hbm_bytes_per_launch") if (tr.get("evaluations") == evals_total and world == 1) else None
        buf = torch.zeros(size, dtype=torch.float64, device="cuda")

        def fused_step():
            bt.evaluate_normal(ps, to_host=False)
            bt.assemble(n_nodes, buf.data_ptr(), zero_first=True)
            if use_dist:
                dist.all_reduce(buf)

        for _ in range(2):
            fused_step()
        torch.cuda.synchronize()
        barrier()
        n_f = max(args.steps, 1)
        ctx.timer_start()
        f0 = time.perf_counter()
        for _ in range(n_f):
            fused_step()
        f_kernel_ms = ctx.timer_stop() / n_f           # evaluate + finalize + assemble (+ all-reduce)
        torch.cuda.synchronize()
        barrier()
        fdt = torch.tensor([time.perf_counter() - f0], dtype=torch.float64, device="cuda")
        # points the kernel actually loads: chunks whose bounding sphere cannot touch the reading
        # grid are skipped without reading them (vgx_reg_batch_count_live); constraints that share a
        # reference submap load the SAME points, which the launch order lets them share in one L2
        lv, uq = bt.count_live(ps, unique=True)
        live = torch.tensor([float(lv), float(uq)], dtype=torch.float64, device="cuda")
        if use_dist:
            dist.all_reduce(fdt, op=dist.ReduceOp.MAX)
            dist.all_reduce(live, op=dist.ReduceOp.SUM)
        fdt, live, unique_pts = float(fdt.item()), float(live[0].item()), float(live[1].item())
        # algorithmic bytes: every DISTINCT loaded point once (20 B) + the 8 neighbours of every
        # evaluation that interpolates (32 B); the per-evaluation pricing of SURVEY.md 8d (52 B / 20 B /
        # 0 B), which re-counts a point for every constraint that reads it, is reported beside it
        alg_bytes = unique_pts * BYTES_NO_CORR_FUSED + corr_total * (BYTES_PER_EVAL_FUSED - BYTES_NO_CORR_FUSED)
        per_eval_bytes = corr_total * BYTES_PER_EVAL_FUSED + max(live - corr_total, 0.0) * BYTES_NO_CORR_FUSED
        out = {"value": evals_total * n_f / fdt / 1e6, "unit": "Mresiduals+Jacobians/s",
               "ms_per_step": fdt / n_f * 1e3, "stream_ms_per_step": f_kernel_ms,
               "kernels": "reg_eval_reduce_lean_kernel + reg_finalize_kernel + reg_assemble_kernel"
                          + (" + RCCL all-reduce" if use_dist else ""),
               "evaluations": evals_total, "with_correspondence": corr_total, "loaded_after_culling": live,
               "distinct_points_loaded": unique_pts,
               "algorithmic_bytes_per_step": alg_bytes,
               "per_evaluation_pricing_bytes_per_step": per_eval_bytes,
               "per_evaluation_pricing_GBs": per_eval_bytes * n_f / fdt / 1e9,
               "pricing": "algorithmic = 20 B x distinct loaded points + 32 B x interpolating evaluations; "
                          "per_evaluation_pricing re-counts a point for every constraint that reads it (52 / 20 / 0 B) "
                          "and can exceed the HBM peak where constraints share points through the L2",
               "algorithmic_GBs": alg_bytes * n_f / fdt / 1e9,
               # NOT an HBM fraction: algorithmic bytes include neighbour bytes the L2 / MALL serve
               "algorithmic_over_hbm_peak": alg_bytes * n_f / fdt / 1e9 / HBM_PEAK_GBS,
               # counter bytes (profiles/hbm_traffic.json, a separate rocprofv3 --pmc run of this workload)
               # / this run's time / 8 TB/s: the HBM fraction proper
               "traffic_from_profiles": tr_bytes,
               "hbm_frac": (tr_bytes / (f_kernel_ms * 1e-3) / 1e9 / HBM_PEAK_GBS) if tr_bytes else None,
               "value_with_correspondence": corr_total * n_f / fdt / 1e6,
               "allreduce_bytes": int(size * 8) if use_dist else 0,
               "cost": float(buf[0].item()), "cost_vs_materialised": None}
        if world == 1 and ref_cost is not None:
            out["cost_vs_materialised"] = abs(out["cost"] - ref_cost) / max(ref_cost, 1e-30)
        return out

    fused = None
    fused_fo = None
    if not args.no_fused:
        fused = fused_bench(batch, poses, n_sub, n_con, with_corr_total, total_evals, checksum, "config3")
        if fo:
            fo_cost = float(residuals[:fo["R"]].double().pow(2).sum().item()) if world == 1 else None
            fused_fo = fused_bench(fo["batch"], fo["poses"], len(fo["poses"]), fo["n"],
                                   fo_out["with_corr_total"], fo_out["R_total"], fo_cost, "full_overlap")

    # ---- the reference's SHIPPED configuration (voxgraph_mapper.yaml:34-35): explicit_to_implicit =
    # isosurface points, sampling_ratio 0.05, both directions of every pair (pose_graph.cpp:62-71),
    # ESDF distance (registration_cost_function.h:35) -- one solver evaluation = one fused batched pass;
    # the per-point-set std::mt19937 streams are generated on the device (mt_generate_kernel)
    shipped = None
    if not args.no_shipped and not args.no_fused:
        cfg_s = capi.default_config(registration_point_type=capi.POINTS_ISOSURFACE, sampling_ratio=0.05)
        shard = lpt_shards([n_iso[a] + n_iso[b] for a, b in pairs], world)[rank]
        pairs_s = [(int(a), int(b)) for c in shard for a, b in (pairs[c], pairs[c][::-1])]
        gidx_s = [2 * c + k for c in shard for k in (0, 1)]
        buf_s = torch.zeros(capi.fused_size(n_sub, 2 * n_con), dtype=torch.float64, device="cuda")

        def shipped_eval(ctx_s, submaps_s):
            cfs_s = [capi.RegistrationCostFunction(ctx_s, submaps_s[a], submaps_s[b], cfg_s) for a, b in pairs_s]
            batch_s = capi.RegistrationBatch(ctx_s, cfs_s, pairs_s, global_index=gidx_s, n_global=2 * n_con)

            def shipped_step():
                batch_s.evaluate_normal(poses, to_host=False)
                batch_s.assemble(n_sub, buf_s.data_ptr(), zero_first=True)
                if use_dist:
                    dist.all_reduce(buf_s)

            for _ in range(2):
                shipped_step()
            torch.cuda.synchronize()
            barrier()
            n_s = max(args.steps, 1)
            ctx_s.timer_start()
            s0 = time.perf_counter()
            for _ in range(n_s):
                shipped_step()
            stream_ms = ctx_s.timer_stop() / n_s
            torch.cuda.synchronize()
            barrier()
            sdt = torch.tensor([time.perf_counter() - s0], dtype=torch.float64, device="cuda")
            rs = torch.tensor([float(batch_s.num_residuals())], dtype=torch.float64, device="cuda")
            if use_dist:
                dist.all_reduce(sdt, op=dist.ReduceOp.MAX)
                dist.all_reduce(rs, op=dist.ReduceOp.SUM)
            e = {"residuals_per_evaluation": float(rs.item()),
                 "ms_per_evaluation": float(sdt.item()) / n_s * 1e3,
                 "stream_ms_per_evaluation": stream_ms,
                 "Mresiduals_per_s": float(rs.item()) * n_s / float(sdt.item()) / 1e6,
                 "cost": float(buf_s[0].item())}
            for o in [batch_s] + cfs_s:
                o.destroy()
            return e
        shipped = {"config": "registration_method explicit_to_implicit (isosurface points), sampling_ratio 0.05, "
                             "mirrored constraints, ESDF distance (voxgraph_mapper.yaml:34-35, pose_graph.cpp:62-71)",
                   "constraints": 2 * n_con, "isosurface_points_per_submap": float(np.mean(n_iso)),
                   "brick_layout": "apron (the headline's submaps)",
                   "what": "one solver evaluation: device mt19937 streams + fused normal equations of every "
                           "constraint + assembly" + (" + RCCL all-reduce" if use_dist else "")}
        shipped.update(shipped_eval(ctx, submaps))
        tr = (PROFILE_TRAFFIC.get("fused") or {}).get("shipped") or {}
        tr_bytes = tr.get("hbm_bytes_per_launch") if (tr.get("evaluations") == shipped["residuals_per_evaluation"]
                                                     and world == 1) else None
        shipped["traffic_from_profiles"] = tr_bytes
        shipped["fused_kernel_ms_from_profiles"] = tr.get("avg_ms_rocprof") if tr_bytes else None
        shipped["hbm_frac"] = (tr_bytes / (tr["avg_ms_rocprof"] * 1e-3) / 1e9 / HBM_PEAK_GBS) \
            if (tr_bytes and tr.get("avg_ms_rocprof")) else None
        shipped["hbm_frac_note"] = "fused kernel alone: counter bytes / its rocprofv3 average duration / 8 TB/s (profiles/)"
        if not args.no_quad:
            # what a sampling session would configure: vgx_ctx_set_brick_layout(VGX_BRICKS_QUAD) -- the same
            # submaps with a 2x2x2 neighbourhood in 32 contiguous bytes (scattered evaluations are bound by the
            # 64-byte lines they touch); same draws, same results
            ctx_q = capi.Context(local_rank)
            ctx_q.set_stream(stream.cuda_stream)
            ctx_q.set_brick_layout(capi.BRICKS_QUAD)
            subs_q = []
            for k in range(n_sub):
                sm = capi.Submap.synth_city(ctx_q, k, args.voxel_size, 16, args.block_min, args.block_dims,
                                            args.truncation, args.esdf_max, 10.0, true_poses[k], args.seed)
                sm.extract_isosurface_points(1.0)
                sm.release_raw_layers()
                subs_q.append(sm)
            q = shipped_eval(ctx_q, subs_q)
            q["brick_layout"] = "quad (vgx_ctx_set_brick_layout(VGX_BRICKS_QUAD): 4.25 x the grid memory)"
            trq = (PROFILE_TRAFFIC.get("fused") or {}).get("shipped_quad") or {}
            okq = trq.get("evaluations") == q["residuals_per_evaluation"] and world == 1 and trq.get("avg_ms_rocprof")
            q["traffic_from_profiles"] = trq.get("hbm_bytes_per_launch") if okq else None
            q["hbm_frac"] = (trq["hbm_bytes_per_launch"] / (trq["avg_ms_rocprof"] * 1e-3) / 1e9 / HBM_PEAK_GBS) if okq else None
            q["cost_equals_apron"] = bool(q["cost"] == shipped["cost"])
            shipped["quad_bricks"] = q
            for sm in subs_q:
                sm.destroy()
            ctx_q.close()

    # ---- the in-process multi-GPU component (vgx_reg_multi_*: one process, one vgx_ctx + host thread per
    # GPU, fixed-order sum over xGMI peer mappings) -- the PRODUCT's multi-GPU path (voxgraph is one process).
    #   N = 1 (default)          : two contexts on this one GPU: what the threads, events and the sum cost
    #   --inprocess --gpus N     : one process, contexts on devices 0..N-1
    #   N > 1 under torchrun     : after the per-rank measurements every rank waits at a barrier while
    #                              rank 0 drives all N GPUs through the component, so the driver's
    #                              scaling runs time it beside the one-process-per-GPU RCCL route
    multi_ctx = None
    if not args.no_fused and not args.no_multi_ctx:
        barrier()
        if rank == 0:
            if world > 1:
                devices = [0] * world if dryrun else list(range(world))
            elif args.inprocess:
                devices = [0] * args.gpus if dryrun else list(range(args.gpus))
            else:
                devices = [local_rank, local_rank]
            try:
                multi_ctx = multi_context_bench(capi, ctx, torch, args, devices, submaps, true_poses, pairs, weights_bytes,
                                                poses, cfg, batch if world == 1 else None, n_sub, n_con)
            except Exception as e:       # an optional section must never cost the line (peer access, memory, ...)
                multi_ctx = {"error": repr(e), "device_ids": devices}
            finally:
                # the library selects each context's device for its calls; PyTorch must find the rank's own
                # device current again (its allocator and the RCCL communicator live there)
                torch.cuda.set_device(local_rank)
        barrier()

    # ---- metric 2: full pose-graph solve (harness LM, stand-in for ceres::Solve) ---
    solve = None
    if not args.no_solve:
        from harness import lm
        from harness.backends import GpuBackend
        info = [1.0, 1.0, 2500.0, 2500.0]                       # voxgraph_mapper.yaml:41-47
        edges = [lm.RelativePoseEdge.from_poses(k, k + 1, poses[k], poses[k + 1], info)
                 for k in range(n_sub - 1)]
        backend = GpuBackend(capi, ctx, batch, n_sub, dist if use_dist else None)
        backend(poses)                                           # warm

        def rmse(p):
            # gauge: submap 0 is fixed at its true pose
            return float(np.sqrt(((p[:, :3] - true_poses[:, :3]) ** 2).sum(1).mean()))

        def timed_solve(**kw):
            # one untimed solve first (imports, index tables, allocator warm-up), like
            # the warm-up steps of the headline loop
            lm.solve(lm.Problem(backend, n_sub, pairs, edges), poses, max_seconds=1e9, **kw)
            torch.cuda.synchronize()
            barrier()
            s0 = time.perf_counter()
            prob = lm.Problem(backend, n_sub, pairs, edges)
            # no wall-clock stop rule here: every rank must take the same number of
            # evaluations (each one is a collective)
            x, summ = lm.solve(prob, poses, max_seconds=1e9, **kw)
            torch.cuda.synchronize()
            barrier()
            sdt = torch.tensor([time.perf_counter() - s0], dtype=torch.float64, device="cuda")
            if use_dist:
                dist.all_reduce(sdt, op=dist.ReduceOp.MAX)
            return {"ms": float(sdt.item()) * 1e3, "iterations": summ["iterations"],
                    "gpu_evaluation_ms": summ["backend_seconds"] * 1e3,
                    "host_linear_algebra_ms": summ["host_linear_algebra_seconds"] * 1e3,
                    "evaluations": summ["evaluations"], "termination": summ["termination"],
                    "initial_cost": summ["initial_cost"], "final_cost": summ["final_cost"],
                    "position_rmse_m_before": rmse(poses), "position_rmse_m_after": rmse(x)}

        # (a) to convergence: Ceres' default function_tolerance 1e-6 decides
        solve = timed_solve(parameter_tolerance=1e-10)
        solve["stop_rule"] = "function_tolerance 1e-6 (Ceres default), parameter_tolerance off"
        # (b) the reference's exact rule: Ceres stops when |step| <= 3e-3 (|x| + 3e-3)
        #     (pose_graph.cpp:93); |x| is ~4 km here, so it fires on the first step
        solve["reference_stop_rule"] = timed_solve(parameter_tolerance=3e-3)
        solve["reference_stop_rule"]["stop_rule"] = "parameter_tolerance 3e-3 relative to |x| (pose_graph.cpp:93)"
        solve["solver"] = "harness/lm.py (LM, banded Cholesky on the host; Ceres absent)"
        solve["split"] = ("gpu_evaluation_ms = time inside the registration backend (fused pass, assembly, copy of "
                          "the fused buffer, all-reduce): the product; host_linear_algebra_ms = the harness' own "
                          "assembly + banded Cholesky, which Ceres does in the real system")

    out = None
    if rank == 0:
        passes = args.steps * args.inner
        value = total_evals * passes / dt / 1e6
        # roofline of the dominant kernel (reg_eval_points_kernel<16,float,4>) on rank 0
        bytes_contract = R * BYTES_PER_EVAL
        # `achieved` prices what this launch has to move (DESIGN.md "Roofline accounting"): 36 B written
        # per evaluation, 20 B per registration point the kernel reads (none for the tiles whose chunks
        # all miss the reading submap's block box; once per group of constraints where the launch order
        # lets them share a reference submap's points in one L2, else once per constraint), and 32 B of
        # neighbours per evaluation that interpolates.  Chunk-granular count: conservative, the kernel
        # culls whole 1024-point tiles.
        bytes_conservative = R * BYTES_OUT + pr["read"] * BYTES_POINT + with_corr * BYTES_NEIGHBOURS
        achieved = bytes_conservative / (kernel_ms * 1e-3) / 1e9
        traffic = traffic_fo = traffic_fo_plain = None
        if PROFILE_TRAFFIC:
            t = PROFILE_TRAFFIC
            # same workload (per-launch residual count) as the PMC passes were taken on
            if t.get("residuals_per_launch") == R and t.get("n_gpus") == world:
                traffic = t.get("hbm_bytes_per_launch")
            tf = t.get("full_overlap") or {}
            if fo and tf.get("residuals_per_launch") == fo["R"] and t.get("n_gpus") == world:
                traffic_fo = tf.get("hbm_bytes_per_launch")
            tp = t.get("full_overlap_plain_order") or {}
            if fo and tp.get("residuals_per_launch") == fo["R"] and t.get("n_gpus") == world:
                traffic_fo_plain = tp.get("hbm_bytes_per_launch")
        out = {
            "metric": "Mresiduals+Jacobians/s per GPU; full pose-graph solve ms (200 submaps)",
            "value": value, "unit": "Mresiduals+Jacobians/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "inprocess_gpus": args.gpus if args.inprocess else None,
            "ms_per_step": dt / args.steps * 1e3, "higher_is_better": True,
            "scaling": "strong", "vs_baseline": None, "dtype": "f32",
            "data": "synthetic" if not dryrun else f"synthetic (DRY RUN: all ranks on one GPU, {dryrun})",
            "config": {"workload": f"configs[2]: synthetic {n_sub} submaps @ "
                                   f"{args.block_dims[0] * 16}x{args.block_dims[1] * 16}x{args.block_dims[2] * 16} voxels "
                                   f"({args.voxel_size} m), {n_con} overlap constraints, kVoxels points, "
                                   "all points (sampling_ratio -1), residual + both Jacobians",
                       "submaps": n_sub, "constraints": n_con,
                       "residuals_per_pass": int(total_evals),
                       "passes_per_step": args.inner,
                       "step": f"{args.inner} consecutive passes over all {n_con} constraints "
                               "(one batched launch per pass per rank)",
                       "parallelism": f"pair-sharded x{world} (LPT" + (" on bytes moved at the initial poses" if world > 1 else "")
                                      + "), submaps replicated",
                       "point_order": "extraction (block, then voxel linear index)",
                       "solve_stop_rule": "solve.ms: Ceres-default function_tolerance 1e-6 (NOT the reference's "
                                          "rule); solve.reference_stop_rule: parameter_tolerance 3e-3 "
                                          "(pose_graph.cpp:93), which fires on the first step of this graph"},
            "value_per_gpu": value / world,
            "value_with_correspondence": with_corr_total * passes / dt / 1e6,
            "ms_per_pass": dt / passes * 1e3,
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS, "traffic": traffic,
                         # `traffic` is REPLAYED from profiles/hbm_traffic.json (rocprofv3 --pmc passes of this
                         # same workload, collected by profiles/collect.sh), not measured in this run
                         "traffic_from_profiles": traffic,
                         "traffic_source": PROFILE_TRAFFIC.get("source") if traffic else None,
                         "hbm_frac": (traffic / (kernel_ms * 1e-3) / 1e9 / HBM_PEAK_GBS) if traffic else None,
                         "hbm_frac_note": "counter bytes / this run's kernel time / 8 TB/s",
                         # SURVEY.md 8(d)'s literal pricing, 88 B x every evaluation: NOT an HBM fraction here
                         # (exceeds 1): most evaluations of this workload find no reading block (20 B + 36 B) and
                         # culled tiles are written without being read (36 B) -- `frac` prices those as such;
                         # roofline_full_overlap.plain_order is the workload the 88 B figure describes
                         "contract_88B_frac": bytes_contract / (kernel_ms * 1e-3) / 1e9 / HBM_PEAK_GBS,
                         "contract_88B_frac_note": "not an HBM fraction: prices bytes this launch does not move",
                         "kernel": "reg_eval_points_kernel<16,float,4>",
                         "kernel_ms": kernel_ms, "kernel_ms_max_over_ranks": kernel_ms_max,
                         "bytes_per_unit": BYTES_PER_EVAL, "bytes_per_unit_without_correspondence": BYTES_NO_CORR,
                         "bytes_per_unit_culled": BYTES_OUT,
                         "units_per_launch": int(R),
                         "bytes_per_launch": int(bytes_conservative),
                         "points_read": pr,
                         "pricing": "36 B out per evaluation + 20 B per registration point read (tiles whose "
                                    "512-point chunks all miss the reading submap's block box are written as "
                                    "zeros without reading their points) + 32 B of neighbours per evaluation "
                                    "that interpolates; see roofline_full_overlap for the workload the 88 B "
                                    "contract figure describes",
                         "all_88B_would_read_GBs": bytes_contract / (kernel_ms * 1e-3) / 1e9,
                         "traffic_GBs": (traffic / (kernel_ms * 1e-3) / 1e9) if traffic else None,
                         "with_correspondence_frac": with_corr / max(R, 1)},
            "fused": fused,
            "shipped_config": shipped,
            "multi_context": multi_ctx,
            "solve": solve,
            "setup_s": setup_s,
            "residual_checksum": checksum,
        }
        if fo_out:
            fk = fo_out["kernel_ms"] * 1e-3
            fp = fo_out["points"]
            fo_bytes = fo_out["R"] * BYTES_OUT + fp["read"] * BYTES_POINT + fo_out["with_corr"] * BYTES_NEIGHBOURS
            plain = None
            if fo_out.get("plain_kernel_ms"):
                pk = fo_out["plain_kernel_ms"] * 1e-3
                plain = {"what": "the same constraints launched in plain constraint-major order "
                                 "(VGX_POINTS_TILE_ORDER=0): no two concurrent constraints share a submap, every "
                                 "evaluation moves its own 88 B -- the regime SURVEY.md 8d's contract figure describes",
                         "kernel_ms": fo_out["plain_kernel_ms"], "kernel_ms_max_over_ranks": fo_out["plain_kernel_ms_max"],
                         "bytes_per_unit": BYTES_PER_EVAL,
                         "achieved": fo_out["R"] * BYTES_PER_EVAL / pk / 1e9,
                         "frac": fo_out["R"] * BYTES_PER_EVAL / pk / 1e9 / HBM_PEAK_GBS,
                         "traffic": traffic_fo_plain, "traffic_from_profiles": traffic_fo_plain,
                         "hbm_frac": (traffic_fo_plain / pk / 1e9 / HBM_PEAK_GBS) if traffic_fo_plain else None,
                         "traffic_GBs": (traffic_fo_plain / pk / 1e9) if traffic_fo_plain else None}
            out["roofline_full_overlap"] = {
                "workload": f"the same {n_sub} submaps, each registered against {args.overlap_copies} duplicates of "
                            f"itself perturbed by N(0, {args.pose_sigma} m) / N(0, {args.yaw_sigma} rad) "
                            f"({fo['n']} constraints)",
                "bound": "hbm", "kernel": "reg_eval_points_kernel<16,float,4>",
                "kernel_ms": fo_out["kernel_ms"], "kernel_ms_max_over_ranks": fo_out["kernel_ms_max"],
                "units_per_launch": int(fo_out["R"]), "bytes_per_unit": BYTES_PER_EVAL,
                "points_read": fp, "bytes_per_launch": int(fo_bytes),
                "pricing": "as roofline: 36 B out per evaluation + 20 B per registration point READ (the launch "
                           "order runs the constraints of a reference submap side by side on one XCD, its points "
                           "are fetched once per group) + 32 B per interpolating evaluation",
                "achieved": fo_bytes / fk / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": fo_bytes / fk / 1e9 / HBM_PEAK_GBS,
                "frac_note": "algorithmic bytes / time / 8 TB/s; part of the neighbour bytes is served by the L2: "
                             "hbm_frac is the HBM fraction proper",
                "traffic_from_profiles": traffic_fo,
                "hbm_frac": (traffic_fo / fk / 1e9 / HBM_PEAK_GBS) if traffic_fo else None,
                # the contract's per-evaluation pricing re-counts a point for every constraint that reads it
                # and can exceed the HBM peak where the L2 serves the repeats; plain_order is where it applies
                "per_evaluation_pricing_GBs": fo_out["R"] * BYTES_PER_EVAL / fk / 1e9,
                "plain_order": plain,
                "with_correspondence_frac": fo_out["with_corr"] / max(fo_out["R"], 1),
                "traffic": traffic_fo,
                "traffic_GBs": (traffic_fo / fk / 1e9) if traffic_fo else None,
                "value": fo_out["R_total"] * fo_out["passes"] / fo_out["dt"] / 1e6,
                "value_with_correspondence": fo_out["with_corr_total"] * fo_out["passes"] / fo_out["dt"] / 1e6,
                "unit_value": "Mresiduals+Jacobians/s",
                "fused": fused_fo}
    # CPU baseline: rank 0, N = 1 only (bounded sample)
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        out["cpu_baseline"] = cpu_baseline(capi, ctx, args, true_poses, poses, pairs,
                                           args.cpu_seconds)
        # metric 2 on the CPU, DERIVED (not run: it would take minutes): the converged solve's
        # evaluation count x this graph's residuals / the measured CPU evaluation rates
        if solve:
            cb = out["cpu_baseline"]
            work = solve["evaluations"] * total_evals / 1e6          # M evaluations in the solve
            est = {"evaluations": solve["evaluations"],
                   "port_all_cores_s": work / cb["value"], "port_4_threads_s": work / cb["value_4_threads"],
                   "note": "derived from the measured rates above; excludes the host linear algebra"}
            rs = cb.get("reference_source") or {}
            if rs.get("value_4_threads"):
                est["reference_source_4_threads_s"] = work / rs["value_4_threads"]
            cb["solve_estimate"] = est
    elif rank == 0:
        out["cpu_baseline"] = None
    # second hot path (does not shard: replicas only) -- rank 0, N = 1
    if rank == 0 and world == 1 and not args.no_tsdf:
        out["tsdf"] = tsdf_bench(capi, ctx, torch)
        out["finish_submap"] = finish_bench(capi, ctx, args, true_poses)
    # config 5 (every rank takes part: its constraints are sharded like config 3's)
    if not args.no_config5:
        for o in [batch] + cfs + ([fo["batch"]] + cfs_fo if fo else []) + submaps:
            o.destroy()                                            # make room: config 5 brings its own 1000 submaps
        del residuals, jac_ref, jac_read
        torch.cuda.empty_cache()
        c5 = config5_bench(capi, ctx, torch, dist, use_dist, rank, world, args)
        if rank == 0:
            out["config5"] = c5
            if c5.get("parity"):
                parity_entries.append(c5["parity"])
    if rank == 0 and world == 1 and not args.no_config2:
        from harness import pipeline
        # 30 submaps, 10 Hz, 10 s per submap = 100 scans per submap; the
        # default line carries a bounded cut of the same session (10 submaps x 30 scans)
        full = args.pipeline
        out["pipeline_config2"] = pipeline.run(capi, ctx, torch, n_submaps=30 if full else 10,
                                               scans_per_submap=100 if full else 30)
        out["pipeline_config2"]["cut"] = "full: 30 submaps x 100 scans" if full else \
            "bounded: 10 submaps x 30 scans of the 30 x 100 session (python bench.py --pipeline runs all of it)"
    if rank == 0:
        if not args.no_parity:
            from harness import parity_gate
            out["parity"] = parity_gate.merge(parity_entries)
        print(json.dumps(out))
    if use_dist:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
