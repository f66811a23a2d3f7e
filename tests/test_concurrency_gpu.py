"""Integrate while optimising -- the reference's defining overlap (voxgraph_mapper.cpp:218-238: optimizePoseGraph runs on a
std::async thread while the ROS thread keeps integrating scans; invariant :464-471: finished submaps are immutable).

Thread A integrates a session of scans into the ACTIVE layer (reproducible mode: the result is defined bit for bit) while
thread B evaluates the pose graph's registration constraints on FINISHED submaps of the SAME context -- fused passes and
drop-in Evaluate calls.  The two sides have their own stream and lock (include/voxgraph_amd.h "THREADING AND STREAMS"):
both results must equal the serial run's bit for bit, nothing may deadlock, and the two together must take less wall
clock than one after the other."""
import threading
import time

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
F = np.float32


@pytest.fixture(scope="module")
def capi():
    from voxgraph_amd import capi
    capi.load()
    return capi


def _scans(n_scans, n_az=256, n_el=24):
    """a LiDAR-like sensor walking through a 10 x 8 x 4 m room; no axis-aligned rays (azimuths offset by a third of a step)"""
    az = np.linspace(-np.pi, np.pi, n_az, endpoint=False) + (2 * np.pi / n_az) / 3.0
    el = np.linspace(-0.3, 0.3, n_el) + 0.004
    A, E = np.meshgrid(az, el)
    d = np.stack([np.cos(E) * np.cos(A), np.cos(E) * np.sin(A), np.sin(E)], -1).reshape(-1, 3)
    out = []
    for k in range(n_scans):
        origin = np.array([-2.0 + 0.02 * k, 0.5 - 0.005 * k, 0.3 + 0.001 * k])
        lo, hi = np.array([-5.0, -4.0, -1.0]) - origin, np.array([5.0, 4.0, 3.0]) - origin
        with np.errstate(divide="ignore", invalid="ignore"):
            t = np.where(d > 0, hi / d, np.where(d < 0, lo / d, np.inf)).min(1)
        out.append((np.array([1, 0, 0, 0, *origin], F), (d * t[:, None]).astype(F)))
    return out


def _graph(capi, ctx, n_sub=6):
    """finished submaps of the analytic city and the registration constraints between neighbours"""
    from harness.bench_common import BYTES_PER_EVAL  # noqa: F401  (the harness is importable: same scene as bench.py)
    poses = np.array([[6.4 * k, 0.4 * (k % 2), 0.0, 0.02 * k] for k in range(n_sub)])
    subs = []
    for k in range(n_sub):
        sm = capi.Submap.synth_city(ctx, k, 0.2, 16, (-4, -4, -2), (8, 8, 4), 0.6, 2.0, 10.0, poses[k], 2)
        sm.extract_voxel_points(1.0, 0.3, True)
        sm.release_raw_layers()
        subs.append(sm)
    pairs = np.array([(a, a + 1) for a in range(n_sub - 1)] + [(a, a + 2) for a in range(n_sub - 2)], np.int32)
    cfg = capi.default_config(registration_point_type=capi.POINTS_VOXELS)
    cfs = [capi.RegistrationCostFunction(ctx, subs[a], subs[b], cfg) for a, b in pairs]
    batch = capi.RegistrationBatch(ctx, cfs, pairs)
    guess = poses + np.array([0.05, -0.03, 0.01, 0.004]) * (np.arange(n_sub)[:, None] % 3)
    return subs, cfs, batch, pairs, guess


def _evaluate(capi, batch, cfs, pairs, guess, n_normal, n_dropin):
    """thread B's work: n_normal fused passes (the per-constraint [45] blocks to the host) and n_dropin drop-in Evaluates"""
    blocks = []
    for it in range(n_normal):
        p = guess + 1e-3 * it
        status, host = batch.evaluate_normal(p, to_host=True)
        assert not status.any()
        blocks.append(host.copy())
    rows = []
    for it in range(n_dropin):
        c = it % len(cfs)
        a, b = pairs[c]
        n = cfs[c].num_residuals()
        r, jo, je = np.zeros(n), np.zeros((n, 4)), np.zeros((n, 4))
        assert cfs[c].Evaluate([guess[a], guess[b]], r, [jo, je])
        rows.append((r, jo, je))
    return blocks, rows


def _integrate(capi, ctx, scans):
    """thread A's work: the whole session into a fresh layer, reproducible mode; returns the layer's download"""
    layer = capi.TsdfLayer(ctx, 0.2, 16)
    integ = capi.FastTsdfIntegrator(ctx, capi.voxgraph_tsdf_config(deterministic=1), layer)
    for T, pts in scans:
        integ.integratePointCloud(T, pts)
    out = layer.download()
    dropped = layer.stats()[1]
    integ.destroy()
    layer.destroy()
    return out, dropped


def test_scans_integrate_while_the_pose_graph_is_evaluated(capi):
    ctx = capi.Context(0)
    try:
        assert ctx.get_tsdf_stream() != ctx.get_stream() and ctx.get_tsdf_stream() != 0
        scans = _scans(200)
        subs, cfs, batch, pairs, guess = _graph(capi, ctx)
        n_normal, n_dropin = 300, 10
        # warm both sides (allocations, first-use initialisation)
        _integrate(capi, ctx, scans[:3])
        _evaluate(capi, batch, cfs, pairs, guess, 2, 2)
        ctx.synchronize()
        # ---- one after the other
        t0 = time.perf_counter()
        layer_serial, dropped = _integrate(capi, ctx, scans)
        t_a = time.perf_counter() - t0
        assert dropped == 0
        t0 = time.perf_counter()
        blocks_serial, rows_serial = _evaluate(capi, batch, cfs, pairs, guess, n_normal, n_dropin)
        t_b = time.perf_counter() - t0
        # ---- together, on two threads (twice: the bits must match both times, the wall clock is the better of the two --
        # a first concurrent run has been seen at 0.93 of the serial sum on a box where the next ones took 0.71 and 0.76)
        t_both_best = None
        for attempt in range(2):
            res, err, took = {}, [], {}

            def run(name, fn):
                try:
                    s0 = time.perf_counter()
                    res[name] = fn()
                    took[name] = time.perf_counter() - s0
                except BaseException as e:    # noqa: BLE001  (a failed assertion in a thread must fail the test)
                    err.append((name, repr(e)))
            ta = threading.Thread(target=run, args=("layer", lambda: _integrate(capi, ctx, scans)))
            tb = threading.Thread(target=run, args=("reg", lambda: _evaluate(capi, batch, cfs, pairs, guess, n_normal, n_dropin)))
            t0 = time.perf_counter()
            ta.start()
            tb.start()
            ta.join(timeout=300)
            tb.join(timeout=300)
            t_both = time.perf_counter() - t0
            assert not ta.is_alive() and not tb.is_alive(), "deadlock: a thread did not finish within 300 s"
            assert not err, err
            # ---- the same bits
            (layer_conc, dropped_c), (blocks_conc, rows_conc) = res["layer"], res["reg"]
            assert dropped_c == 0
            for x, y in zip(layer_serial, layer_conc):
                assert np.array_equal(x.view(np.uint8), y.view(np.uint8))
            assert len(layer_serial[0]) > 20
            for x, y in zip(blocks_serial, blocks_conc):
                assert np.array_equal(x.view(np.uint64), y.view(np.uint64))
            for (r0, a0, b0), (r1, a1, b1) in zip(rows_serial, rows_conc):
                assert np.array_equal(r0, r1) and np.array_equal(a0, a1) and np.array_equal(b0, b1)
            assert any(np.abs(b).max() > 0 for b in blocks_serial)
            print(f"serial: scans {t_a * 1e3:.1f} ms + evaluations {t_b * 1e3:.1f} ms = {(t_a + t_b) * 1e3:.1f} ms; "
                  f"concurrent: {t_both * 1e3:.1f} ms (scans {took['layer'] * 1e3:.1f}, evaluations {took['reg'] * 1e3:.1f})")
            t_both_best = t_both if t_both_best is None else min(t_both_best, t_both)
        # ---- and an overlap: less than one after the other (the reproducible mode waits for the device several times
        # per scan -- the other side's kernels run meanwhile)
        assert t_both_best < 0.97 * (t_a + t_b), (t_a, t_b, t_both_best)
        for o in [batch] + cfs + subs:
            o.destroy()
    finally:
        ctx.close()


def test_racing_scans_under_a_running_solve_are_legal_interleavings_and_get_in(capi):
    """The DEFAULT mode under the reference's defining overlap: racing scans -- one launch each -- on
    thread A while thread B loops fused solver evaluations on the same context.  Every scan's event log is replayed
    through the oracle (tests/test_tsdf_replay_gpu.py's checker: 0 violations allowed), the solver's blocks stay
    bit-identical to the serial run, and -- the fused passes run five workgroups deep while the context has an
    integrator -- a scan submitted under the solver completes well inside the fused kernel's duration."""
    from oracle import pyoracle as orc
    import torch
    ctx = capi.Context(0)
    try:
        subs, cfs, batch, pairs, guess = _graph(capi, ctx, n_sub=8)
        scans = _scans(24, n_az=1024, n_el=64)
        ocfg, gcfg = orc.voxgraph_tsdf_config(), capi.voxgraph_tsdf_config()
        layer = capi.TsdfLayer(ctx, 0.2, 16)
        integ = capi.FastTsdfIntegrator(ctx, gcfg, layer)
        _, serial = batch.evaluate_normal(guess, to_host=True)
        stop, err, done = threading.Event(), [], [0]

        def solver():
            try:
                while not stop.is_set():
                    _, blocks = batch.evaluate_normal(guess, to_host=True)
                    assert np.array_equal(blocks.view(np.uint64), serial.view(np.uint64))
                    done[0] += 1
            except BaseException as e:    # noqa: BLE001
                err.append(repr(e))
        th = threading.Thread(target=solver)
        th.start()
        time.sleep(0.05)
        try:
            # ---- replayed scans (the logging instantiation of the shipped kernel), the solver running
            integ.set_event_trace(8 << 20)
            totals = dict(scans=0, exchanges=0, updates=0, overrun=0)
            for T, pts in scans[:12]:
                s0, o0, _ = integ.download_sets()
                l0 = layer.download()
                n_upd = integ.integratePointCloud(T, pts)
                trace, lost = integ.read_event_trace()
                s1, o1, (off_s, off_o, _) = integ.download_sets()
                rep = orc.tsdf_replay_check(ocfg, 0.2, 16, T, pts, None, False, (off_s, off_o), (s0, o0), (s1, o1), l0,
                                            layer.download(), trace)
                assert lost == 0 and rep["errors"] == 0, rep["first_error"]
                assert rep["required_updates"] == n_upd
                totals["scans"] += 1
                totals["exchanges"] += rep["observed_exchanges"]
                totals["updates"] += n_upd
                totals["overrun"] += rep["overrun_exchanges"]
            integ.set_event_trace(0)
            # ---- latency of the shipped (unlogged) kernel under the solver: submit -> complete on the TSDF stream alone
            dev = [torch.from_numpy(p).cuda() for _, p in scans]
            torch.cuda.synchronize()
            lat = []
            for k, (T, pts) in enumerate(scans):
                time.sleep(0.004)
                t0 = time.perf_counter()
                integ.integrate_device(T, dev[k].data_ptr(), None, len(pts))
                ctx.synchronize_tsdf()
                lat.append((time.perf_counter() - t0) * 1e6)
        finally:
            stop.set()
            th.join(timeout=120)
        assert not err, err
        assert done[0] > 10 and layer.stats()[1] == 0
        lat = np.sort(np.array(lat))
        print(f"racing scans under a running solve: {totals}; {done[0]} solver evaluations meanwhile; scan latency p50 "
              f"{np.median(lat):.0f} us, max {lat[-1]:.0f} us")
        # (a generous bar: the probe measures 0.2-0.4 ms; before the fused passes left room it was 0.7 ms median on the big
        # graph -- here the solver's kernel is short, so the bar only says "does not wait for whole evaluations")
        assert np.median(lat) < 2000
        for o in [integ, layer, batch] + cfs + subs:
            o.destroy()
    finally:
        ctx.close()


def test_finish_submap_hands_over_across_the_two_streams(capi):
    """vgx_submap_from_tsdf_layer right behind the last scan, no host synchronisation in between: the registration side's
    kernels (brick building, ESDF, point extraction) must see the layer as the TSDF stream left it"""
    ctx = capi.Context(0)
    try:
        scans = _scans(6)

        def build(sync):
            layer = capi.TsdfLayer(ctx, 0.2, 16)
            integ = capi.FastTsdfIntegrator(ctx, capi.voxgraph_tsdf_config(deterministic=1), layer)
            import torch
            dev = [torch.from_numpy(p).cuda() for _, p in scans]
            torch.cuda.synchronize()
            for (T, p), d in zip(scans, dev):
                integ.integrate_device(T, d.data_ptr(), None, len(p))        # asynchronous
            if sync:
                ctx.synchronize()
            sm = capi.Submap.from_tsdf_layer(ctx, layer, 7)
            sm.generate_esdf()
            n = sm.extract_voxel_points(1.0, 0.3, True)
            xyz, dist, w = sm.download_points(capi.POINTS_VOXELS)
            for o in (sm, integ, layer):
                o.destroy()
            return n, xyz, dist, w
        n0, x0, d0, w0 = build(True)
        n1, x1, d1, w1 = build(False)
        assert n0 == n1 > 100
        assert np.array_equal(x0, x1) and np.array_equal(d0, d1) and np.array_equal(w0, w1)
    finally:
        ctx.close()


def test_scans_produced_on_another_stream_are_ordered_by_tsdf_wait_for_stream(capi):
    """vgx_ctx_tsdf_wait_for_stream: scan points PRODUCED on another stream (here PyTorch's: a long matrix
    product queued in front of the copy, so that the points land late) and integrated on the TSDF stream with no host
    synchronisation in between -- the device-side wait must order the integration behind the producer: the same layer as
    with a synchronise after every scan"""
    import torch
    ctx = capi.Context(0)
    try:
        scans = _scans(6, n_az=512, n_el=32)
        pinned = [torch.from_numpy(p).pin_memory() for _, p in scans]
        ballast = torch.randn((4096, 4096), device="cuda")
        producer = torch.cuda.Stream()                                  # (a named stream: NULL means "the registration stream")

        def run(sync_every_scan):
            layer = capi.TsdfLayer(ctx, 0.2, 16)
            integ = capi.FastTsdfIntegrator(ctx, capi.voxgraph_tsdf_config(deterministic=1), layer)
            bufs = [torch.zeros((len(p), 3), dtype=torch.float32, device="cuda") for _, p in scans]
            torch.cuda.synchronize()
            ctx.synchronize()
            for k, (T, pts) in enumerate(scans):
                with torch.cuda.stream(producer):
                    for _ in range(4):
                        ballast @ ballast                               # a few milliseconds of work in front of the copy
                    bufs[k].copy_(pinned[k], non_blocking=True)
                if sync_every_scan:
                    producer.synchronize()
                else:
                    ctx.tsdf_wait_for_stream(producer.cuda_stream)      # the TSDF stream waits, on the device
                integ.integrate_device(T, bufs[k].data_ptr(), None, len(pts))
            out = layer.download()
            for o in (integ, layer):
                o.destroy()
            return out
        want, got = run(True), run(False)
        assert len(want[0]) > 20
        for x, y in zip(want, got):
            assert np.array_equal(x.view(np.uint8), y.view(np.uint8))
    finally:
        ctx.close()
