"""The replay checker itself (oracle/tsdf_replay.c), on the CPU: it must ACCEPT the event log of the oracle's own
single-thread run -- one legal interleaving of voxblox::FastTsdfIntegrator::integratePointCloud's steps (the call of
voxgraph/src/frontend/measurement_processors/pointcloud_integrator.cpp:83) -- and REJECT that log once a single event is
dropped, duplicated or altered: the one-in-10^5 lost or repeated update a statistical comparison cannot see.
tests/test_tsdf_replay_gpu.py holds the racing GPU kernel's logs to the same checker."""
import numpy as np
import pytest

from oracle import pyoracle as orc

F = np.float32


def _room_scan(n_az, n_el, origin, phase=0.0):
    az = np.linspace(-np.pi, np.pi, n_az, endpoint=False) + phase
    el = np.linspace(-0.35, 0.35, n_el)
    A, E = np.meshgrid(az, el)
    d = np.stack([np.cos(E) * np.cos(A), np.cos(E) * np.sin(A), np.sin(E)], -1).reshape(-1, 3)
    lo, hi = np.array([-5.0, -4.0, -1.0]) - origin, np.array([5.0, 4.0, 3.0]) - origin
    with np.errstate(divide="ignore"):
        t = np.where(d > 0, hi / d, lo / d).min(1)
    return (d * t[:, None]).astype(F)


def _events(trace):
    """[(offset, length, kind)] of a log"""
    out, i = [], 0
    while i < len(trace):
        kind = int(trace[i]) & 0xff
        n = 4 if kind != 5 else 6 + ((int(trace[i]) >> 8) & 0xffffffff)
        out.append((i, n, kind))
        i += n
    return out


def _session(logged):
    """three scans of a room from a moving sensor, shipped yaml, colours; the state before / after scan `logged` and its log"""
    vs, vps = 0.2, 16
    cfg = orc.voxgraph_tsdf_config()
    layer = orc.TsdfLayer(vs, vps)
    integ = orc.FastTsdfIntegrator(cfg, layer)
    rng = np.random.default_rng(5)
    scans = []
    for k in range(3):
        origin = np.array([0.3 * k, -0.2 * k, 0.05 * k])
        pts = _room_scan(512, 32, origin, 0.001 * (k + 1))
        pts[rng.integers(0, len(pts), 40)] *= F(0.001)          # too short: invalid
        pts[rng.integers(0, len(pts), 40)] *= F(5.0)            # too long: clearing rays
        col = rng.integers(0, 256, (len(pts), 4)).astype(np.uint8)
        T = np.array([1, 0, 0, 0, *origin], F)
        scans.append((T, pts, col))
    for T, pts, col in scans[:logged]:
        integ.integratePointCloud(T, pts, col)
    sets_pre = integ.download_sets()[:2]
    layer_pre = layer.download()
    integ.set_log(8 << 20)
    T, pts, col = scans[logged]
    updates = integ.integratePointCloud(T, pts, col)
    trace, lost = integ.read_log()
    assert lost == 0 and len(trace) > 0
    a, b, offsets = integ.download_sets()
    return dict(cfg=cfg, vs=vs, vps=vps, T=T, pts=pts, col=col, offsets=offsets, sets_pre=sets_pre, sets_post=(a, b),
                layer_pre=layer_pre, layer_post=layer.download(), trace=trace, updates=updates)


@pytest.fixture(scope="module")
def session():
    return _session(2)


def _check(s, trace=None, **over):
    a = dict(s, **over)
    return orc.tsdf_replay_check(a["cfg"], a["vs"], a["vps"], a["T"], a["pts"], a["col"], False, a["offsets"], a["sets_pre"],
                                 a["sets_post"], a["layer_pre"], a["layer_post"], a["trace"] if trace is None else trace)


def test_the_single_threads_own_log_is_a_legal_interleaving(session):
    rep = _check(session)
    assert rep["errors"] == 0, rep["first_error"]
    assert rep["required_updates"] == session["updates"] == rep["fold_records"] == rep["fold_events"]
    assert rep["overrun_exchanges"] == 0 and rep["start_skips"] == 0
    assert rep["rays_cast"] > 500 and rep["rays_stopped_early"] > 100
    assert rep["valid_points"] < rep["points"] and rep["colour_writes"] > 100 and rep["voxels_with_several_links"] > 100
    print(rep)
    # the FIRST scan of the session: an empty layer (every block new), rays that walk all the way to the sensor
    first = _session(0)
    rep = _check(first)
    assert rep["errors"] == 0, rep["first_error"]
    assert rep["required_updates"] == first["updates"] and rep["new_blocks"] > 50 and rep["rays_walked_to_end"] > 10
    print(rep)


def _drop(trace, ev):
    return np.concatenate([trace[:ev[0]], trace[ev[0] + ev[1]:]])


def test_tampered_logs_are_rejected(session):
    trace = session["trace"]
    ev = _events(trace)
    folds = [e for e in ev if e[2] == 5]
    obs = [e for e in ev if e[2] == 4]
    starts = [e for e in ev if e[2] == 1]
    rng = np.random.default_rng(11)
    tried = 0

    def rejected(t, what, **over):
        nonlocal tried
        tried += 1
        rep = _check(session, trace=t, **over)
        assert rep["errors"] > 0, what
        return rep["first_error"]

    # one voxel update dropped / applied twice
    for k in rng.integers(0, len(folds), 6):
        msg = rejected(_drop(trace, folds[k]), "a dropped fold")
        assert "one path" in msg or "never applied" in msg or "without a logged fold" in msg, msg
        f = folds[k]
        msg = rejected(np.concatenate([trace, trace[f[0]:f[0] + f[1]]]), "a duplicated fold")
        assert "twice" in msg, msg
    # one bit of a published word
    for k in rng.integers(0, len(folds), 6):
        t = trace.copy()
        t[folds[k][0] + 3] ^= np.uint64(1 << int(rng.integers(0, 23)))
        rejected(t, "a flipped bit in a published word")
    # an update credited to another point's ray
    for k in rng.integers(0, len(folds), 6):
        t = trace.copy()
        t[folds[k][0] + 6] += np.uint64(1)
        rejected(t, "a record renamed")
    # an observed-set exchange dropped; one whose returned value is altered (the ray's decisions no longer follow, or the
    # slot's path breaks); a start-set exchange whose returned value says "present" where it was not
    for k in rng.integers(0, len(obs), 6):
        rejected(_drop(trace, obs[k]), "a dropped observed-set exchange")
        t = trace.copy()
        t[obs[k][0] + 3] = t[obs[k][0] + 2] if t[obs[k][0] + 3] != t[obs[k][0] + 2] else np.uint64(12345)
        rejected(t, "an altered observed-set exchange")
    for k in rng.integers(0, len(starts), 6):
        t = trace.copy()
        t[starts[k][0] + 3] = t[starts[k][0] + 2] if t[starts[k][0] + 3] != t[starts[k][0] + 2] else np.uint64(12345)
        rejected(t, "an altered start-set exchange")
    # the state after the scan: one voxel's distance, one colour byte, one set slot
    bi, d, w, c = (x.copy() for x in session["layer_post"])
    touched = np.argwhere(w > 0)
    b, v = touched[len(touched) // 2]
    d2 = d.copy(); d2[b, v] = np.nextafter(d2[b, v], F(10))
    rejected(trace, "a voxel that is not what the log leaves", layer_post=(bi, d2, w, c))
    c2 = c.copy(); c2[b, v, 1] ^= 1
    rejected(trace, "a colour that is not what the log leaves", layer_post=(bi, d, w, c2))
    s2 = session["sets_post"][1].copy(); s2[int(np.flatnonzero(s2)[7])] += np.uint64(1)
    rejected(trace, "a set slot that is not what the log leaves", sets_post=(session["sets_post"][0], s2))
    # a point moved by one ulp: the oracle's ray is no longer the logged one (values, voxels or sdf differ)
    cast = [int(trace[e[0] + 1]) for e in ev if e[2] == 3]
    p2 = session["pts"].copy()
    p2[cast[len(cast) // 2]] *= F(1.0001)
    rejected(trace, "a log of other points", pts=p2)
    assert tried >= 40


def test_an_exchange_behind_the_stop_is_counted_not_hidden(session):
    """the kernel's stated liberty: an exchange behind a ray's stop is legal only as a counted overrun, and only with the
    value the ray's next voxel gives"""
    trace = session["trace"]
    ev = _events(trace)
    # a ray that stopped early: its last observed event is the stop; append the exchange of the step behind it, on a slot
    # whose path we extend consistently (the sequential log never has such an event)
    by_point = {}
    for e in ev:
        if e[2] == 4:
            by_point.setdefault(int(trace[e[0] + 1]), []).append(e)
    totals = {int(trace[e[0] + 1]): int(trace[e[0] + 2]) for e in ev if e[2] == 3}
    p = next(p for p, es in by_point.items() if len(es) < totals[p])
    rep0 = _check(session)
    # a fabricated exchange for step len(es) with a wrong value is rejected (not the oracle's voxel)
    step = len(by_point[p])
    t = np.concatenate([trace, np.array([4 | (step << 8), p, 999, 999], np.uint64)])
    rep = _check(session, trace=t)
    assert rep["errors"] > 0 and "oracle's voxel gives" in rep["first_error"], rep["first_error"]
    assert rep0["errors"] == 0
