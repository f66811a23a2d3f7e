"""bench.py's TSDF block (metric 3) and the finishSubmap() timings."""
import os
import sys
import time

import numpy as np

from harness.bench_common import HBM_PEAK_GBS, effective_cores  # noqa: F401

def _room_points(dirs, origin):
    """first hit of unit rays from `origin` with the inside of a 10 x 8 x 4 m room"""
    lo, hi = np.array([-5.0, -4.0, -1.0]) - origin, np.array([5.0, 4.0, 3.0]) - origin
    with np.errstate(divide="ignore", invalid="ignore"):
        t = np.where(dirs > 0, hi / dirs, np.where(dirs < 0, lo / dirs, np.inf))
    return (dirs * t.min(1)[:, None]).astype(np.float32)


def tsdf_sources_sha():
    """sha256 over the TSDF sources the sort-based paths are made of (what profiles/tsdf_launches.sh records with a trace)"""
    import hashlib
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    h = hashlib.sha256()
    for name in ("vgx_tsdf.hip", "vgx_tsdf_det.hip", "vgx_tsdf_internal.h"):
        try:
            h.update(open(os.path.join(root, "voxgraph_amd", "csrc", name), "rb").read())
        except OSError:
            return None
    return h.hexdigest()[:16]


def profiled_launches():
    """kernel launches per scan of the sort-based paths, from the committed rocprofv3 trace of the same sensor shapes
    (profiles/r06_tsdf_launches.txt, made by profiles/tsdf_launches.sh): {("fast" | "merged", "lidar" | "rgbd"): launches}
    -- and only if the trace was taken on THESE sources (the script records their hash; ADVICE r4: a stale count divided
    into a fresh time is a wrong microseconds-per-launch without notice).  Returns (counts, note)."""
    import re
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    path = os.path.join(root, "profiles", "r06_tsdf_launches.txt")
    out, key, sha = {}, None, None
    try:
        for line in open(path):
            m = re.match(r"sources sha256: (\w+)", line)
            if m:
                sha = m.group(1)
            m = re.match(r"=== (fast|merged) det=\d (lidar|rgbd)", line)
            if m:
                key = (m.group(1), m.group(2))
            m = re.match(r"(\d+) launches, period", line)
            if m and key:
                out[key] = int(m.group(1))
                key = None
    except OSError:
        return {}, "no committed trace (profiles/r06_tsdf_launches.txt)"
    now = tsdf_sources_sha()
    if sha is None or now is None or sha != now:
        return {}, f"the committed trace was taken on other sources (trace {sha}, these {now}): launch counts dropped"
    return out, f"profiles/r06_tsdf_launches.txt, sources {sha}"


def sensor_cases():
    """the two sensor shapes of BASELINE.json: name -> (unit ray directions, voxel size, integrator settings, block box)"""
    u, v = np.meshgrid((np.arange(640) - 319.5) / 525.0, (np.arange(480) - 239.5) / 525.0)
    d_rgbd = np.stack([np.ones_like(u), -u, -v], -1).reshape(-1, 3)
    d_rgbd /= np.linalg.norm(d_rgbd, axis=1, keepdims=True)
    az = np.linspace(-np.pi, np.pi, 1024, endpoint=False)
    el = np.deg2rad(np.linspace(-16.6, 16.6, 64))
    A, E = np.meshgrid(az, el)
    d_lidar = np.stack([np.cos(E) * np.cos(A), np.cos(E) * np.sin(A), np.sin(E)], -1).reshape(-1, 3)
    return {
        "rgbd_640x480_0.05m": (d_rgbd, 0.05, dict(default_truncation_distance=0.15, max_ray_length_m=5.0),
                               (-8, -6, -2), (16, 12, 7)),
        "lidar_64x1024_0.20m_voxgraph_yaml": (d_lidar, 0.20, dict(
            default_truncation_distance=0.60, max_ray_length_m=16.0, use_const_weight=1,
            use_weight_dropoff=1, use_sparsity_compensation_factor=1,
            sparsity_compensation_factor=20.0), (-3, -3, -2), (6, 6, 4)),
    }


def session_scans(dirs, scans):
    """`scans` sensor poses along a short path through the room and the clouds seen from them (sensor frame)"""
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
    return poses, clouds


AGE_PASSES = 4   # untimed sessions before the timed one (steady from ~80 scans: profiles/r05_tsdf_age.txt)


def tsdf_bench(capi, ctx, torch, scans=20, cpu_scans=8, solver_step=None, solver_ms_alone=None):
    """Second hot path (TSDF): whole scans resident in HBM, one integratePointCloud per
    scan into the active layer, HIP-event timed.  Two sensor shapes from BASELINE.json:
    RGB-D 640x480 @ 0.05 m voxels (config 4) and OS1-64-shaped LiDAR 64x1024 @ 0.20 m with
    the shipped yaml (config 2's integrator settings)."""
    from oracle import pyoracle as orc
    out = {}
    cases = sensor_cases()
    launches, launches_note = profiled_launches()
    for name, (dirs, vs, kw, bmin, bdim) in cases.items():
        sensor = "rgbd" if name.startswith("rgbd") else "lidar"
        poses, clouds = session_scans(dirs, scans)
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
        updates = 0
        ctx.timer_start()
        for k in range(1, scans):
            integ.integrate_device(poses[k], dev[k].data_ptr(), None, n_pts)
        ms_fresh = ctx.timer_stop()
        # The integrator of a mapping session is OLD: the reference creates one FastTsdfIntegrator and points it at every
        # new submap's layer (pointcloud_integrator.cpp:66-75), and voxblox's ApproxHashSet never forgets between its
        # 10 000-scan resets -- a mark left by scan N at slot (h + N) reads as "present" for the voxel with hash h - k in
        # scan N + k, so rays of later scans are cut short by marks of earlier ones (the oracle and the reproducible mode
        # carry the artefact bit for bit).  A fresh integrator's first ~80 scans therefore do more work per scan than all
        # the scans after them (profiles/r05_tsdf_age.txt: LiDAR kernel 45 -> 24 us, reproducible mode 0.31 -> 0.25 ms).
        # `ms_per_scan` is the steady state: the session integrated AGE_PASSES times untimed, then once more into a
        # fresh layer, timed; the fresh integrator's figure is reported beside it.
        for _ in range(AGE_PASSES):
            for k in range(scans):
                integ.integrate_device(poses[k], dev[k].data_ptr(), None, n_pts)
        # (three timed sessions, each into its own fresh layer, the interpreter's collector off: a session is 0.4-1 ms of
        # device time, and one host hiccup between two launches has been seen to add a third to it; the median is reported)
        import gc
        integrator_age = (1 + AGE_PASSES) * scans
        session_ms, grew, layer_s = [], 0, None
        gc.collect()
        gc.disable()
        for _ in range(3):
            if layer_s is not None:
                layer_s.destroy()
            layer_s = new_layer()
            integ.setLayer(layer_s)
            integ.integrate_device(poses[0], dev[0].data_ptr(), None, n_pts)        # warm-up scan: the layer's blocks
            ctx.synchronize()
            g0 = layer_s.growths()
            ctx.timer_start()
            for k in range(1, scans):
                integ.integrate_device(poses[k], dev[k].data_ptr(), None, n_pts)
            session_ms.append(ctx.timer_stop())
            grew = max(grew, layer_s.growths() - g0)
        gc.enable()
        ms = float(np.median(session_ms))
        # what a scan costs the mapping thread while the pose graph is being optimised on the same context
        # (voxgraph_mapper.cpp:218-238): racing scans at the sensor's cadence, thread B looping solver evaluations
        latency = None
        if solver_step is not None:
            from harness.bench_latency import latency_block
            lay_l = new_layer()
            integ.setLayer(lay_l)
            hz, n_lat = (30.0, 90) if sensor == "rgbd" else (10.0, 40)   # 3 s / 4 s of sensor time (p99 of 45 scans was its maximum)
            try:
                latency = latency_block(capi, ctx, torch, integ, poses, dev, n_pts, hz, n_lat, solver_step, solver_ms_alone)
            except Exception as e:   # noqa: BLE001
                latency = {"error": repr(e)[:300]}
            ctx.synchronize()
            integ.setLayer(layer_s)
            lay_l.destroy()
        # second pass: voxel updates per scan (the count needs a sync per scan) and, with the stream
        # drained around every launch, the duration of each scan's kernel by itself (HIP events)
        layer2 = new_layer()
        integ.setLayer(layer2)
        # (the interpreter's garbage collector off meanwhile: a generation-2 collection between timer_start and the
        # launch -- 35-40 ms, seen in round 5 at the same scan of every run -- is the harness, not the kernel; the
        # MEDIAN of the scans is reported, mean / min / max beside it)
        import gc
        gc.collect()
        gc.disable()
        per_scan_ms = []
        for k in range(scans):
            ctx.synchronize()
            ctx.timer_start()
            integ.integrate_device(poses[k], dev[k].data_ptr(), None, n_pts)
            per_scan_ms.append(ctx.timer_stop())
        gc.enable()
        kernel_ms = float(np.median(per_scan_ms[1:]))
        # the floor of that measurement: a one-point scan timed the same way (launch + an almost empty kernel)
        one_point_ms = []
        for k in range(1, min(6, scans)):
            ctx.synchronize()
            ctx.timer_start()
            integ.integrate_device(poses[k], dev[k].data_ptr(), None, 1)
            one_point_ms.append(ctx.timer_stop())
        # an ORGANISED cloud (vgx_tsdf_integrator_set_cloud_width: sensor_msgs/PointCloud2.width): 16 x 16 tiles of beams
        width = 640 if sensor == "rgbd" else 1024
        layer2c = new_layer()
        integ.setLayer(layer2c)
        integ.set_cloud_width(width)
        gc.disable()
        org_ms = []
        for k in range(scans):
            ctx.synchronize()
            ctx.timer_start()
            integ.integrate_device(poses[k], dev[k].data_ptr(), None, n_pts)
            org_ms.append(ctx.timer_stop())
        layer2d = new_layer()
        integ.setLayer(layer2d)
        integ.integrate_device(poses[0], dev[0].data_ptr(), None, n_pts)
        ctx.synchronize()
        ctx.timer_start()
        for k in range(1, scans):
            integ.integrate_device(poses[k], dev[k].data_ptr(), None, n_pts)
        org_b2b_ms = ctx.timer_stop() / (scans - 1)
        gc.enable()
        integ.set_cloud_width(0)
        for o in (layer2c, layer2d):
            o.destroy()
        layer2b = new_layer()
        integ.setLayer(layer2b)
        walks = []
        for k in range(scans):
            u_ = integ.integrate_device(poses[k], dev[k].data_ptr(), None, n_pts, count=True)
            updates += u_ if k >= 1 else 0
            if k >= 1:
                # what this scan's rays did (vgx_tsdf_integrator_walk_stats) + its voxel updates
                walks.append(dict(integ.walk_stats(), updates=u_))
        n_blocks, dropped = layer_s.stats()
        # The two ceilings of a kernel made of device-scope atomics on scattered 8-byte words (DESIGN.md 3 "TSDF
        # latency model"), measured on this GPU with the operation by itself -- a chain of dependent exchanges on
        # an 8 MiB table (an approximate hash set):
        #   * one wavefront alone: the round trip a ray waits for once per voxel step (it learns whether to take
        #     step k + 1 only when the exchange of step k has come back),
        #   * 8192 wavefronts side by side: what the memory system sustains, in operations per second.
        rt_unloaded_ns = capi.atomic_roundtrip_ns(ctx, 8 << 20, 1, 2000)
        sat_waves = 8192
        rt_saturated_ns = capi.atomic_roundtrip_ns(ctx, 8 << 20, sat_waves, 100)
        atomic_peak_gops = sat_waves * 64 / rt_saturated_ns
        v1 = os.environ.get("VGX_TSDF_KERNEL") == "v1"
        longest = float(np.mean([w_["longest_chain"] for w_ in walks]))
        if v1:
            # one-thread-per-point kernel: one exchange per voxel step + per update a load and a CAS of {distance,
            # weight} + a load and a CAS of the colour where it blends; the chain is the longest ray's steps
            ops_scan = float(np.mean([w_["exchanges"] + 2 * w_["updates"] + 2 * w_["colour_blends"] for w_ in walks]))
            chain_trips = longest
        else:
            # cooperative kernel (vgx_tsdf_coop.hip): exchanges + peeks of the walk; per distinct voxel and workgroup one
            # block-table load, one load each of {distance, weight} and colour, one CAS (+ one of the colour where a
            # record blends: counted as one per fold, an upper bound) + the folds that had to be repeated.  The chain a
            # scan cannot be shorter than: the point load and the start-set exchange, the longest ray's rounds, then
            # table load -> voxel load -> CAS -> colour CAS.
            ops_scan = float(np.mean([w_["exchanges"] + w_["peeks"] + 5 * w_["voxel_folds"] + w_["cas_retries"] for w_ in walks]))
            chain_trips = 2.0 + longest + 4.0
        chain_ms = chain_trips * rt_unloaded_ns * 1e-6
        throughput_ms = ops_scan / atomic_peak_gops * 1e-6
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
        def det_session():
            integ7.integrate_device(poses[0], dev[0].data_ptr(), None, n_pts)
            ctx.synchronize()
            d0 = time.perf_counter()
            for k in range(1, scans):
                integ7.integrate_device(poses[k], dev[k].data_ptr(), None, n_pts)
            ctx.synchronize()
            return (time.perf_counter() - d0) * 1e3 / (scans - 1)
        det_ms_fresh = det_session()
        for _ in range(AGE_PASSES - 1):      # (the mode's layer does not depend on how old the integrator's sets are
            for k in range(scans):           #  beyond what the oracle's does: tests/test_tsdf_deterministic_gpu.py)
                integ7.integrate_device(poses[k], dev[k].data_ptr(), None, n_pts)
        det_sessions = []
        layer7s = None
        for _ in range(3):                   # (three timed sessions, each into its own fresh layer: the median is reported)
            if layer7s is not None:
                layer7s.destroy()
            layer7s = new_layer()
            integ7.setLayer(layer7s)
            det_sessions.append(det_session())
        det_ms = float(np.median(det_sessions))
        det_updates = integ7.integrate_device(poses[1], dev[1].data_ptr(), None, n_pts, count=True)
        for o in (integ7, layer7, layer7s):
            o.destroy()
        # the drop-in call itself: host pointers (pageable), PCIe upload included, returns when done;
        # layer created the way voxblox creates one (no reservation at all)
        layer5 = capi.TsdfLayer(ctx, vs, 16)
        integ5 = capi.FastTsdfIntegrator(ctx, capi.tsdf_config(**kw), layer5)
        integ5.integratePointCloud(poses[0], clouds[0], count=False)
        ctx.synchronize_tsdf()
        h0 = time.perf_counter()
        for k in range(1, scans):
            integ5.integratePointCloud(poses[k], clouds[k], count=False)     # (n_updates == NULL: voxblox's void call)
        host_call_ms = (time.perf_counter() - h0) * 1e3 / (scans - 1)       # what the mapping thread spends per scan
        ctx.synchronize_tsdf()
        host_ms = (time.perf_counter() - h0) * 1e3 / (scans - 1)            # ... and the session's rate, last scan done
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
        # ... and in integration_order_mode "sorted" (voxgraph_mapper.yaml:29), three scans
        ol_s = orc.TsdfLayer(vs, 16)
        oi_s = orc.FastTsdfIntegrator(orc.tsdf_config(integration_order=2, **kw), ol_s)
        layer9 = capi.TsdfLayer(ctx, vs, 16)
        integ9 = capi.FastTsdfIntegrator(ctx, capi.tsdf_config(deterministic=1, integration_order=capi.TSDF_ORDER_SORTED, **kw), layer9)
        cu_s = gu_s = 0
        for k in range(3):
            cu_s += oi_s.integratePointCloud(poses[k], clouds[k])
            gu_s += integ9.integrate_device(poses[k], dev[k].data_ptr(), None, n_pts, count=True)
        sbi, sd, sw, sc = ol_s.download()
        tbi, td_, tw_, tc = layer9.download()
        det_parity_sorted = {"scans": 3, "voxel_updates_equal": bool(gu_s == cu_s), "blocks": int(len(sbi)),
                             "bit_identical": bool(np.array_equal(sbi, tbi) and np.array_equal(sd.view(np.uint32), td_.view(np.uint32))
                                                   and np.array_equal(sw.view(np.uint32), tw_.view(np.uint32)) and np.array_equal(sc, tc)),
                             "differs_from_mixed_order": bool(not (np.array_equal(sbi, obi[:len(sbi)]) and len(sbi) == len(obi)
                                                                   and np.array_equal(sd.view(np.uint32), od.view(np.uint32)))) if cpu_scans == 2 else None}
        for o in (integ9, layer9):
            o.destroy()
        # the same port on all host cores: the path does not shard (one active submap), so this is
        # REPLICAS -- one integrator + layer per thread, every thread the same scans.  All replicas start
        # behind a barrier and run their whole sequence inside ONE foreign call (oracle/tsdf_oracle.c
        # orc_tsdf_integrate_sequence: no interpreter lock between scans); rate = total points / wall clock
        # from the barrier's release to the LAST replica's finish.
        import threading
        cores, cores_info = effective_cores()
        one_scan_s = cdt / cpu_scans
        seq_scans = min(scans - 1, 8)
        repeats = max(int(np.ceil(50 / seq_scans)), int(np.ceil(0.4 / max(one_scan_s * seq_scans, 1e-6))))
        seq_poses = np.stack(poses[1:1 + seq_scans]).astype(np.float32)
        seq_clouds = np.stack(clouds[1:1 + seq_scans]).astype(np.float32)
        age_repeats = int(np.ceil((1 + AGE_PASSES) * scans / seq_scans))
        gate = threading.Barrier(cores)
        t_begin, t_end = [0.0] * cores, [0.0] * cores

        def replica(j):
            # every replica builds and warms its own integrator + layer on its own thread, so that its pages are
            # first touched where it runs
            l_ = orc.TsdfLayer(vs, 16)
            i_ = orc.FastTsdfIntegrator(orc.tsdf_config(**kw), l_)
            i_.integrate_sequence(seq_poses, seq_clouds, age_repeats)    # untimed: the layer exists, pages are touched, the
            gate.wait()                                                  # integrator is as old as the GPU's (see above)
            t_begin[j] = time.perf_counter()
            i_.integrate_sequence(seq_poses, seq_clouds, repeats)
            t_end[j] = time.perf_counter()
        # the one-core rate the all-cores row is held against: the SAME sequence on one thread (the 8-scan sample
        # above includes the layer's first allocations; this is the steady state the replicas run in)
        l1 = orc.TsdfLayer(vs, 16)
        i1 = orc.FastTsdfIntegrator(orc.tsdf_config(**kw), l1)
        i1.integrate_sequence(seq_poses, seq_clouds, age_repeats)
        t1 = time.perf_counter()
        i1.integrate_sequence(seq_poses, seq_clouds, repeats)
        one_core_steady = n_pts * seq_scans * repeats / (time.perf_counter() - t1) / 1e6
        del i1, l1
        threads = [threading.Thread(target=replica, args=(j,)) for j in range(cores)]
        for t_ in threads:
            t_.start()
        for t_ in threads:
            t_.join()
        wall = max(t_end) - min(t_begin)
        one_core = one_core_steady
        all_rate = n_pts * seq_scans * repeats * cores / wall / 1e6
        cpu_all = {"Mpoints_per_s": all_rate, "cores": cores, "host_cpus": cores_info, "kind": "port", "replicas": cores,
                   "wall_s": wall,
                   "scans_per_replica": seq_scans * repeats,
                   "slowest_replica_s": max(e_ - b_ for b_, e_ in zip(t_begin, t_end)),
                   "fastest_replica_s": min(e_ - b_ for b_, e_ in zip(t_begin, t_end)),
                   "one_core_same_sequence_Mpoints_per_s": one_core_steady,
                   "per_core_over_one_core": all_rate / cores / one_core,
                   "at_most_cores_x_one_core": bool(all_rate <= 1.05 * cores * one_core),
                   "sample": f"{cores} replicas (one integrator + layer per thread) x {seq_scans * repeats} scans "
                             f"({repeats} passes over {seq_scans}), started behind a barrier; total points / wall "
                             "clock to the last finish; the restatement is serial within a scan"}
        timed = scans - 1
        alg_bytes_scan = 16.0 * n_pts + 24.0 * updates / timed
        out[name] = {"points_per_scan": n_pts, "scans_timed": timed, "ms_per_scan": ms / timed,
                     "latency_under_solve_us": latency,
                     "ms_per_scan_fresh_integrator": ms_fresh / timed, "integrator_age_scans": integrator_age,
                     "ms_per_scan_of_each_timed_session": [x / timed for x in session_ms],
                     "protocol": f"one integrator for the session (pointcloud_integrator.cpp:66-75): {integrator_age} scans old "
                                 "when the timed session starts (a fresh layer, one warm-up scan); *_fresh_integrator = its very "
                                 "first session (voxblox's approximate sets cut later scans' rays short: DESIGN.md 3)",
                     "Mpoints_per_s": n_pts * timed / ms / 1e3,
                     "Mvoxel_updates_per_s": updates / ms / 1e3,
                     "voxel_updates_per_scan": updates / timed, "blocks": n_blocks,
                     "dropped_updates": dropped, "layer_enlargements_in_timed_region": grew,
                     "algorithmic_GBs": alg_bytes_scan * timed / ms / 1e6,
                     # The racing kernel is made of scattered device-scope atomics; a scan is 1-8 MB, a microsecond of
                     # HBM time (hbm_frac reads 0.005: the wrong yardstick).  Its two ceilings, measured on this
                     # GPU: the LATENCY chain of the longest ray (dependent exchanges x idle round trip) and the
                     # THROUGHPUT of such operations (all of the scan's / what the memory system sustains).
                     # `peak` = the larger of the two lower bounds on the kernel's time, `frac` = peak / measured.
                     "roofline": {"bound": "latency" if chain_ms >= throughput_ms else "atomic-throughput",
                                  "kernel": "tsdf_integrate_kernel<true> (VGX_TSDF_KERNEL=v1)" if v1 else "tsdf_integrate_coop_kernel<false>",
                                  "kernel_ms": kernel_ms,
                                  "kernel_ms_how": "HIP events around each scan's launch, stream drained before; median of the scans",
                                  "kernel_ms_mean": float(np.mean(per_scan_ms[1:])), "kernel_ms_min": float(np.min(per_scan_ms[1:])),
                                  "kernel_ms_max": float(np.max(per_scan_ms[1:])),
                                  "one_point_scan_ms": float(np.median(one_point_ms)),
                                  "organised_cloud": {"width": width, "kernel_ms": float(np.median(org_ms[1:])),
                                                      "back_to_back_ms_per_scan": org_b2b_ms,
                                                      "what": "the same scans declared organised (16 x 16 tiles of beams per "
                                                              "workgroup: vgx_tsdf_integrator_set_cloud_width)"},
                                  "longest_walk_steps": longest, "longest_walk_steps_max": int(max(w_["longest_chain"] for w_ in walks)),
                                  "longest_walk_unit": "voxel steps" if v1 else "rounds of the cooperative walk (one round trip each)",
                                  "dependent_round_trips": chain_trips,
                                  "per_scan": {k_: float(np.mean([w_[k_] for w_ in walks])) for k_ in walks[0]},
                                  "roundtrip_ns_unloaded": rt_unloaded_ns, "latency_chain_ms": chain_ms,
                                  "memory_operations_per_scan": ops_scan,
                                  "atomic_peak_Gops": atomic_peak_gops, "atomic_peak_how":
                                      f"{sat_waves} wavefronts x 64 chains of dependent exchanges on an 8 MiB table",
                                  "atomic_achieved_Gops": ops_scan / kernel_ms * 1e-6,
                                  "atomic_throughput_ms": throughput_ms,
                                  "achieved": kernel_ms, "peak": max(chain_ms, throughput_ms), "unit": "ms",
                                  "frac": max(chain_ms, throughput_ms) / kernel_ms,
                                  "traffic": None,
                                  "bytes_per_launch": alg_bytes_scan,
                                  "hbm_achieved_GBs": alg_bytes_scan / kernel_ms / 1e6, "hbm_peak_GBs": HBM_PEAK_GBS,
                                  "hbm_frac": alg_bytes_scan / kernel_ms / 1e6 / HBM_PEAK_GBS,
                                  "back_to_back_ms_per_scan": ms / timed,
                                  "back_to_back_over_kernel": (ms / timed) / kernel_ms},
                     "host_pointer_call": {"ms_per_scan": host_ms, "Mpoints_per_s": n_pts / host_ms / 1e3,
                                           "caller_ms_per_scan": host_call_ms,
                                           "layer_enlargements": host_growths,
                                           "note": "vgx_tsdf_integrate(n_updates = NULL) into an unreserved layer: pageable host "
                                                   "points read through the integrator's pinned staging, PCIe upload and "
                                                   "enlargements included; ms_per_scan = the session's wall time to the last "
                                                   "scan's completion / scans, caller_ms_per_scan = what the calls themselves took"},
                     "merged_integrator": {"ms_per_scan": merged_ms, "Mpoints_per_s": n_pts / merged_ms / 1e3,
                                           # launch / read-back bound: ~30 (LiDAR) to ~50 (depth image) small kernels -- two
                                                           # stable sorts of rocprim's (7-9 launches each at these sizes) -- and one host
                                                           # read-back per scan (profiles/r04_tsdf_launches.txt); the HBM figure is kept for
                                                           # SURVEY 8d's pricing and is NOT what bounds it
                                                           "roofline": {"bound": "launch", "unit": "GB/s", "hbm_peak_GBs": HBM_PEAK_GBS,
                                                        "launches_per_scan_from_profiles": launches.get(("merged", sensor)),
                                                        "launches_source": launches_note,
                                                        "us_per_launch": (merged_ms * 1e3 / launches[("merged", sensor)]
                                                                          if ("merged", sensor) in launches else None),
                                                        "bytes_per_launch": 16.0 * n_pts + 24.0 * merged_updates,
                                                        "hbm_achieved_GBs": (16.0 * n_pts + 24.0 * merged_updates) / merged_ms / 1e6,
                                                        "hbm_frac": (16.0 * n_pts + 24.0 * merged_updates) / merged_ms / 1e6 / HBM_PEAK_GBS,
                                                        "time": "back-to-back scans, all kernels of a scan (keys, sort, "
                                                                "heads, rays)"},
                                           "voxel_updates_per_scan": merged_updates, "dropped_updates": merged_dropped,
                                           "Mvoxel_updates_per_s": merged_updates / merged_ms / 1e3,
                                           "note": "vgx_tsdf_integrate_merged_device: key + stable radix sort + group heads + "
                                                   "cooperative merge, then every ray written out, sorted by voxel and applied "
                                                   "voxel by voxel in group order (no early-out: the voxels next to the sensor "
                                                   "take one update per group, a sequential f32 chain)"},
                     "reproducible_mode": {"ms_per_scan": det_ms, "ms_per_scan_fresh_integrator": det_ms_fresh,
                                           "ms_per_scan_of_each_timed_session": det_sessions,
                                           "Mpoints_per_s": n_pts / det_ms / 1e3,
                                           # launch / latency bound like the merged integrator
                                           "roofline": {"bound": "launch",
                                                        "launches_per_scan_from_profiles": launches.get(("fast", sensor)),
                                                        "launches_source": launches_note,
                                                        "us_per_launch": (det_ms * 1e3 / launches[("fast", sensor)]
                                                                          if ("fast", sensor) in launches else None),
                                                        "host_waits_per_scan": "2 (the count, the commit) + 1 per extra attempt, each a "
                                                                               "poll of pinned report words behind a one-workgroup "
                                                                               "kernel; the sweeps report through a pinned word"},
                                           "voxel_updates_per_scan": det_updates,
                                           "over_racing_kernel": det_ms / (ms / timed),
                                           "parity_vs_oracle": det_parity,
                                           "parity_vs_oracle_sorted_order": det_parity_sorted,
                                           "note": "vgx_tsdf_config.deterministic = 1: the single-thread visiting "
                                                   "order resolved in parallel (sort by approximate-set slot, "
                                                   "fixed-point sweeps, ordered per-voxel updates); wall clock incl. "
                                                   "its host synchronisations"},
                     "first_scan": {"ms": first_ms, "voxel_updates": first_updates,
                                    "Mvoxel_updates_per_s": first_updates / first_ms / 1e3,
                                    "algorithmic_GBs": (16.0 * n_pts + 24.0 * first_updates) / first_ms / 1e6},
                     "cpu_baseline": {"Mpoints_per_s": one_core_steady,
                                      "Mpoints_per_s_fresh_integrator": n_pts * cpu_scans / cdt / 1e6,
                                      "Mvoxel_updates_per_s_fresh_integrator": cu / cdt / 1e6, "cores": 1,
                                      "kind": "port", "sample": f"{seq_scans * repeats} scans ({repeats} passes over {seq_scans}) "
                                                                f"through an integrator {age_repeats * seq_scans} scans old, "
                                                                "oracle/tsdf_oracle.c (" + orc.build_flags() + "); "
                                                                f"fresh integrator: its first {cpu_scans} scans",
                                      "all_cores": cpu_all}}
        for o in (integ, layer, layer2, layer_s):
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

