"""bench.py's `tsdf.*.latency_under_solve_us`: what a scan costs the mapping thread while the pose graph is being optimised
on the same context -- the reference's defining overlap (voxgraph_mapper.cpp:218-238: optimizePoseGraph on a std::async
thread, the ROS thread keeps integrating).  Thread A submits racing scans (the default mode, one launch each) at the
sensor's cadence and waits for each on the TSDF stream alone (vgx_ctx_synchronize_tsdf); thread B loops fused solver
evaluations on the registration stream.  Per scan: submit -> complete, with and without thread B."""
import threading
import time

import numpy as np


def _percentiles(us):
    a = np.sort(np.asarray(us, np.float64))
    return {"p50": float(np.percentile(a, 50)), "p90": float(np.percentile(a, 90)), "p99": float(np.percentile(a, 99)), "max": float(a[-1]),
            "scans": int(len(a)), "over_1ms": int((a > 1000.0).sum())}


def scan_latency(capi, ctx, torch, integ, T_list, d_clouds, n_points, hz, n_scans, solver_step=None):
    """n_scans racing scans at `hz`, each timed submit -> complete on the host clock; solver_step: a callable thread B loops
    (None: the scans alone).  Returns (latencies in microseconds, solver evaluations completed meanwhile)."""
    stop = threading.Event()
    evals = [0]
    err = []

    def solver():
        try:
            while not stop.is_set():
                solver_step()
                evals[0] += 1
        except BaseException as e:    # noqa: BLE001
            err.append(repr(e))

    th = None
    if solver_step is not None:
        th = threading.Thread(target=solver)
        th.start()
        time.sleep(0.05)              # the solver is up to speed before the first scan arrives
    lat = []
    period = 1.0 / hz
    t_next = time.perf_counter()
    try:
        for k in range(n_scans):
            now = time.perf_counter()
            if now < t_next:
                time.sleep(t_next - now)
            t_next += period
            j = k % len(T_list)
            t0 = time.perf_counter()
            integ.integrate_device(T_list[j], d_clouds[j].data_ptr(), None, n_points)
            ctx.synchronize_tsdf()
            lat.append((time.perf_counter() - t0) * 1e6)
    finally:
        stop.set()
        if th is not None:
            th.join(timeout=60)
    if err:
        raise RuntimeError("solver thread: " + err[0])
    return lat, evals[0]


def latency_block(capi, ctx, torch, integ, T_list, d_clouds, n_points, hz, n_scans, solver_step, solver_ms_alone):
    alone, _ = scan_latency(capi, ctx, torch, integ, T_list, d_clouds, n_points, hz, n_scans, None)
    t0 = time.perf_counter()
    under, evals = scan_latency(capi, ctx, torch, integ, T_list, d_clouds, n_points, hz, n_scans, solver_step)
    wall = time.perf_counter() - t0
    out = {"cadence_Hz": hz, "alone": _percentiles(alone), **_percentiles(under),
           "solver_evaluations_meanwhile": evals,
           "solver_ms_per_evaluation_meanwhile": (wall * 1e3 / evals) if evals else None,
           "solver_ms_per_evaluation_alone": solver_ms_alone,
           "stream_priorities": ctx.stream_priorities(),
           "what": "host clock, submit -> vgx_ctx_synchronize_tsdf, one racing scan (default mode, device-resident points) per "
                   "period; `alone`: no solver running; p50 / p99 / max: thread B looping fused solver evaluations "
                   "(vgx_reg_batch_evaluate_normal, blocks to the host) of the config-3 graph on the registration stream"}
    return out
