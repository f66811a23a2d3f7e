"""Levenberg-Marquardt over 4-DoF (x, y, z, yaw) submap poses.

Mirrors what the reference asks of Ceres (voxgraph/src/backend/pose_graph.cpp:85-106):
trust-region LM, no robust loss (constraint.h:34), first submap constant
(pose_graph_interface.cpp:30-32), yaw wrapped to [-pi, pi)
(local_parameterization/normalize_angle.h:11-16), stop on
parameter_tolerance = 3e-3 (pose_graph.cpp:93) with Ceres' other default
tolerances.  Relative-pose (odometry / loop-closure) edges restate
relative_pose_cost_function_inl.h:8-70 with analytic Jacobians.

A "backend" evaluates every registration constraint at the given poses and
returns the fused normal-equation buffer of SURVEY.md 8e:
    [cost, J^T r (4 n), diag 4x4 blocks of J^T J (16 n), off-diag blocks (16 m)]
"""
import time

import numpy as np
from scipy.linalg import solveh_banded
from scipy.sparse import coo_matrix
from scipy.sparse.csgraph import reverse_cuthill_mckee

try:
    from threadpoolctl import ThreadpoolController
    _THREADPOOLS = None
except ImportError:                                   # pragma: no cover
    ThreadpoolController = None


def normalize_angle(a):
    two_pi = 2.0 * np.pi
    return a - two_pi * np.floor((a + np.pi) / two_pi)


class RelativePoseEdge:
    """RelativePoseCostFunction: residual = sqrt_info * [R(yaw_A)^T (t_B - t_A) - t_obs,
    normalize(yaw_B - yaw_A - yaw_obs)]."""

    def __init__(self, a, b, t_obs, yaw_obs, information_diag):
        self.a, self.b = int(a), int(b)
        self.t_obs = np.asarray(t_obs, np.float64)
        self.yaw_obs = float(yaw_obs)
        self.sqrt_info = np.sqrt(np.asarray(information_diag, np.float64))

    @staticmethod
    def from_poses(a, b, pose_a, pose_b, information_diag):
        c, s = np.cos(pose_a[3]), np.sin(pose_a[3])
        d = pose_b[:3] - pose_a[:3]
        t = np.array([c * d[0] + s * d[1], -s * d[0] + c * d[1], d[2]])
        return RelativePoseEdge(a, b, t, normalize_angle(pose_b[3] - pose_a[3]), information_diag)

    def evaluate(self, poses):
        pa, pb = poses[self.a], poses[self.b]
        c, s = np.cos(pa[3]), np.sin(pa[3])
        d = pb[:3] - pa[:3]
        r = np.array([c * d[0] + s * d[1] - self.t_obs[0], -s * d[0] + c * d[1] - self.t_obs[1],
                      d[2] - self.t_obs[2], normalize_angle(pb[3] - pa[3] - self.yaw_obs)])
        Ja = np.zeros((4, 4))
        Jb = np.zeros((4, 4))
        Rt = np.array([[c, s, 0.0], [-s, c, 0.0], [0.0, 0.0, 1.0]])
        Ja[:3, :3] = -Rt
        Jb[:3, :3] = Rt
        Ja[0, 3] = -s * d[0] + c * d[1]
        Ja[1, 3] = -c * d[0] - s * d[1]
        Ja[3, 3] = -1.0
        Jb[3, 3] = 1.0
        w = self.sqrt_info
        return r * w, Ja * w[:, None], Jb * w[:, None]


def unpack_fused(buf, n_nodes, pairs):
    """fused buffer -> (cost, g [4n], H [4n,4n]).  Reference implementation; Problem
    uses precomputed scatter indices for the same layout."""
    n = n_nodes
    cost = float(buf[0])
    g = np.array(buf[1:1 + 4 * n])
    H = np.zeros((4 * n, 4 * n))
    diag = buf[1 + 4 * n:1 + 20 * n].reshape(n, 4, 4)
    for i in range(n):
        H[4 * i:4 * i + 4, 4 * i:4 * i + 4] = diag[i]
    off = buf[1 + 20 * n:].reshape(-1, 4, 4)
    for c, (a, b) in enumerate(pairs):
        H[4 * a:4 * a + 4, 4 * b:4 * b + 4] += off[c]
        H[4 * b:4 * b + 4, 4 * a:4 * a + 4] += off[c].T
    return cost, g, H


def _block_index(rows, cols):
    """flat indices of the 4x4 blocks (rows[i], cols[i]) in a [4n,4n] matrix -> [m,4,4] r, c"""
    k = np.arange(4)
    r = (4 * np.asarray(rows)[:, None, None] + k[None, :, None]) + 0 * k[None, None, :]
    c = (4 * np.asarray(cols)[:, None, None] + k[None, None, :]) + 0 * k[None, :, None]
    return r, c


class Problem:
    def __init__(self, backend, n_nodes, pairs, edges=(), constant_nodes=(0,)):
        self.backend = backend
        self.n = n_nodes
        self.pairs = [(int(a), int(b)) for a, b in pairs]
        self.edges = list(edges)
        free = np.ones(4 * n_nodes, bool)
        for k in constant_nodes:
            free[4 * k:4 * k + 4] = False
        self.free = np.where(free)[0]
        self.evaluations = 0
        self.backend_seconds = 0.0      # time inside backend(poses): the product's part of a solve
        n = n_nodes
        pa = np.array([p[0] for p in self.pairs], np.int64).reshape(-1)
        pb = np.array([p[1] for p in self.pairs], np.int64).reshape(-1)
        self._diag_rc = _block_index(np.arange(n), np.arange(n))
        self._off_rc = _block_index(pa, pb)
        if self.edges:
            self._ea = np.array([e.a for e in self.edges])
            self._eb = np.array([e.b for e in self.edges])
            self._et = np.array([e.t_obs for e in self.edges])
            self._eyaw = np.array([e.yaw_obs for e in self.edges])
            self._ew = np.array([e.sqrt_info for e in self.edges])
            self._e_aa = _block_index(self._ea, self._ea)
            self._e_bb = _block_index(self._eb, self._eb)
            self._e_ab = _block_index(self._ea, self._eb)

    # ---- reduced, permuted, banded form used by solve() ---------------------------
    def _prepare_reduced(self, poses=None):
        """Index tables for assembling the free-variable normal equations directly in
        symmetric banded storage (node order: see below)."""
        n = self.n
        free_node = np.zeros(n, bool)
        free_node[np.unique(self.free // 4)] = True
        pa = np.array([p[0] for p in self.pairs] + [e.a for e in self.edges], np.int64)
        pb = np.array([p[1] for p in self.pairs] + [e.b for e in self.edges], np.int64)
        adj = coo_matrix((np.ones(2 * len(pa)), (np.r_[pa, pb], np.r_[pb, pa])), shape=(n, n)).tocsr()
        # the narrowest of three orders: reverse Cuthill-McKee, the node numbering itself, and the
        # nodes sorted along the principal axis of their current positions (a map is a thin slab: on
        # config 3's 20 x 10 grid RCM gives a node bandwidth of 34, the numbering 41, the sweep 12;
        # loop closures join nodes that are far apart in the graph and close in space)
        def node_bandwidth(order_):
            pos = -np.ones(n, np.int64)
            pos[order_] = np.arange(len(order_))
            both = (pos[pa] >= 0) & (pos[pb] >= 0)
            return int(np.abs(pos[pa][both] - pos[pb][both]).max()) if both.any() else 0
        rcm = [int(k) for k in reverse_cuthill_mckee(adj, symmetric_mode=True) if free_node[k]]
        natural = [int(k) for k in range(n) if free_node[k]]
        candidates = [rcm, natural]
        if poses is not None:
            xy = np.asarray(poses, np.float64).reshape(-1, 4)[:, :2]
            axis = np.linalg.svd(xy - xy.mean(0), full_matrices=False)[2][0]
            candidates.append([int(k) for k in np.argsort(xy @ axis, kind="stable") if free_node[k]])
        order = min(candidates, key=node_bandwidth)
        node_pos = -np.ones(n, np.int64)
        node_pos[order] = np.arange(len(order))
        self._nf = 4 * len(order)
        var_pos = -np.ones(4 * n, np.int64)           # full variable index -> reduced position
        for k in range(4):
            var_pos[4 * np.array(order) + k] = 4 * node_pos[order] + k
        self._var_pos = var_pos
        self._perm_full = np.argsort(np.where(var_pos >= 0, var_pos, 4 * n))[:self._nf]

        def blocks(rows, cols):
            r, c = _block_index(rows, cols)
            return var_pos[r].ravel(), var_pos[c].ravel()

        srcs = [blocks(np.arange(n), np.arange(n))]
        qa = np.array([p[0] for p in self.pairs], np.int64)
        qb = np.array([p[1] for p in self.pairs], np.int64)
        srcs += [blocks(qa, qb), blocks(qb, qa)]
        if self.edges:
            srcs += [blocks(self._ea, self._ea), blocks(self._eb, self._eb),
                     blocks(self._ea, self._eb), blocks(self._eb, self._ea)]
        R = np.concatenate([x[0] for x in srcs])
        Cc = np.concatenate([x[1] for x in srcs])
        self._coo_keep = (R >= 0) & (Cc >= 0)
        self._coo_r, self._coo_c = R[self._coo_keep], Cc[self._coo_keep]
        self._u = int(np.abs(self._coo_r - self._coo_c).max()) if len(self._coo_r) else 0
        upper = self._coo_r <= self._coo_c
        self._band_sel = upper
        self._band_flat = (self._u + self._coo_r[upper] - self._coo_c[upper]) * self._nf + self._coo_c[upper]

    def evaluate_reduced(self, poses):
        """-> (0.5 sum r^2, g_f [nf], (ab [u+1, nf] upper banded J^T J, coo values))"""
        if not hasattr(self, "_nf"):
            self._prepare_reduced(poses)
        tb = time.perf_counter()
        buf = np.asarray(self.backend(poses))
        self.backend_seconds += time.perf_counter() - tb
        self.evaluations += 1
        n, m = self.n, len(self.pairs)
        cost = float(buf[0])
        g = np.array(buf[1:1 + 4 * n])
        off = buf[1 + 20 * n:1 + 20 * n + 16 * m].reshape(m, 4, 4)
        vals = [buf[1 + 4 * n:1 + 20 * n], off.ravel(), off.transpose(0, 2, 1).ravel()]
        if self.edges:
            ecost, eg_a, eg_b, aa, bb, ab = self._edge_terms(poses)
            cost += ecost
            np.add.at(g.reshape(-1, 4), self._ea, eg_a)
            np.add.at(g.reshape(-1, 4), self._eb, eg_b)
            vals += [aa.ravel(), bb.ravel(), ab.ravel(), ab.transpose(0, 2, 1).ravel()]
        V = np.concatenate(vals)[self._coo_keep]
        band = np.bincount(self._band_flat, weights=V[self._band_sel],
                           minlength=(self._u + 1) * self._nf).reshape(self._u + 1, self._nf)
        return 0.5 * cost, g[self._perm_full], (band, V)

    def reduced_matvec(self, V, x):
        return np.bincount(self._coo_r, weights=V * x[self._coo_c], minlength=self._nf)

    def _edge_terms(self, poses):
        pa, pb = poses[self._ea], poses[self._eb]
        c, s = np.cos(pa[:, 3]), np.sin(pa[:, 3])
        d = pb[:, :3] - pa[:, :3]
        E = len(self._ea)
        r = np.stack([c * d[:, 0] + s * d[:, 1] - self._et[:, 0],
                      -s * d[:, 0] + c * d[:, 1] - self._et[:, 1],
                      d[:, 2] - self._et[:, 2],
                      normalize_angle(pb[:, 3] - pa[:, 3] - self._eyaw)], 1) * self._ew
        Jb = np.zeros((E, 4, 4))
        Jb[:, 0, 0], Jb[:, 0, 1], Jb[:, 1, 0], Jb[:, 1, 1], Jb[:, 2, 2], Jb[:, 3, 3] = c, s, -s, c, 1, 1
        Ja = -Jb.copy()
        Ja[:, 0, 3] = -s * d[:, 0] + c * d[:, 1]
        Ja[:, 1, 3] = -c * d[:, 0] - s * d[:, 1]
        Ja *= self._ew[:, :, None]
        Jb *= self._ew[:, :, None]
        return (float((r * r).sum()), np.einsum("eji,ej->ei", Ja, r), np.einsum("eji,ej->ei", Jb, r),
                np.einsum("eki,ekj->eij", Ja, Ja), np.einsum("eki,ekj->eij", Jb, Jb),
                np.einsum("eki,ekj->eij", Ja, Jb))

    def _edges(self, poses, g, H):
        """all RelativePoseEdge residuals / Jacobians at once (same arithmetic as
        RelativePoseEdge.evaluate)"""
        pa, pb = poses[self._ea], poses[self._eb]
        c, s = np.cos(pa[:, 3]), np.sin(pa[:, 3])
        d = pb[:, :3] - pa[:, :3]
        E = len(self._ea)
        r = np.stack([c * d[:, 0] + s * d[:, 1] - self._et[:, 0],
                      -s * d[:, 0] + c * d[:, 1] - self._et[:, 1],
                      d[:, 2] - self._et[:, 2],
                      normalize_angle(pb[:, 3] - pa[:, 3] - self._eyaw)], 1) * self._ew
        Jb = np.zeros((E, 4, 4))
        Jb[:, 0, 0], Jb[:, 0, 1], Jb[:, 1, 0], Jb[:, 1, 1], Jb[:, 2, 2], Jb[:, 3, 3] = c, s, -s, c, 1, 1
        Ja = -Jb.copy()
        Ja[:, 0, 3] = -s * d[:, 0] + c * d[:, 1]
        Ja[:, 1, 3] = -c * d[:, 0] - s * d[:, 1]
        Ja *= self._ew[:, :, None]
        Jb *= self._ew[:, :, None]
        np.add.at(g.reshape(-1, 4), self._ea, np.einsum("eji,ej->ei", Ja, r))
        np.add.at(g.reshape(-1, 4), self._eb, np.einsum("eji,ej->ei", Jb, r))
        np.add.at(H, self._e_aa, np.einsum("eki,ekj->eij", Ja, Ja))
        np.add.at(H, self._e_bb, np.einsum("eki,ekj->eij", Jb, Jb))
        ab = np.einsum("eki,ekj->eij", Ja, Jb)
        np.add.at(H, self._e_ab, ab)
        np.add.at(H, (self._e_ab[1].transpose(0, 2, 1), self._e_ab[0].transpose(0, 2, 1)),
                  ab.transpose(0, 2, 1))
        return float((r * r).sum())

    def evaluate(self, poses):
        """0.5 * sum r^2, gradient J^T r, Gauss-Newton Hessian J^T J (registration + edges)."""
        tb = time.perf_counter()
        buf = np.asarray(self.backend(poses))
        self.backend_seconds += time.perf_counter() - tb
        self.evaluations += 1
        n = self.n
        cost = float(buf[0])
        g = np.array(buf[1:1 + 4 * n])
        H = np.zeros((4 * n, 4 * n))
        H[self._diag_rc] = buf[1 + 4 * n:1 + 20 * n].reshape(n, 4, 4)
        off = buf[1 + 20 * n:1 + 20 * n + 16 * len(self.pairs)].reshape(-1, 4, 4)
        np.add.at(H, self._off_rc, off)
        np.add.at(H, (self._off_rc[1].transpose(0, 2, 1), self._off_rc[0].transpose(0, 2, 1)),
                  off.transpose(0, 2, 1))
        if self.edges:
            cost += self._edges(poses, g, H)
        return 0.5 * cost, g, H


def _spd_solve(A, b):
    from scipy.linalg import cho_factor, cho_solve
    return cho_solve(cho_factor(A, lower=True, check_finite=False), b, check_finite=False)


def solve(problem, poses0, parameter_tolerance=3e-3, function_tolerance=1e-6,
          gradient_tolerance=1e-10, max_iterations=50, max_seconds=4.0,
          initial_radius=1e4, verbose=False, blas_threads=1):
    """Returns (poses, summary).  Ceres-style LM: (H + D^2/radius) step, gain-ratio
    acceptance, radius update radius / max(1/3, 1 - (2 rho - 1)^3).
    The banded Cholesky runs on `blas_threads` threads: one.  A band of ~60 over 800 columns has
    nothing to parallelise, and OpenBLAS' fork-join costs more than the factorisation (measured on the
    GPU box: 1.39 ms on all 256 cores, 1.03 ms on 4, the reference's Ceres num_threads
    (pose_graph.cpp:96, which governs residual evaluation there, not the linear solver))."""
    global _THREADPOOLS
    if ThreadpoolController is None:
        return _solve(problem, poses0, parameter_tolerance, function_tolerance,
                      gradient_tolerance, max_iterations, max_seconds, initial_radius, verbose)
    if _THREADPOOLS is None:
        _THREADPOOLS = ThreadpoolController()         # scans the loaded libraries once
    # only ever LOWER a pool: an OpenBLAS that initialised with one thread (OMP_NUM_THREADS=1,
    # which torch.distributed.run exports) segfaults when raised to four afterwards
    limits = {}
    for lib in _THREADPOOLS.info():
        api = lib["user_api"]
        limits[api] = min(limits.get(api, blas_threads), max(1, int(lib["num_threads"])))
    with _THREADPOOLS.limit(limits=limits):
        return _solve(problem, poses0, parameter_tolerance, function_tolerance,
                      gradient_tolerance, max_iterations, max_seconds, initial_radius, verbose)


def _solve(problem, poses0, parameter_tolerance, function_tolerance, gradient_tolerance,
           max_iterations, max_seconds, initial_radius, verbose):
    t0 = time.perf_counter()
    x = np.array(poses0, np.float64).copy()
    cost, gf, (band, V) = problem.evaluate_reduced(x)
    u, perm = problem._u, problem._perm_full
    radius, decrease = float(initial_radius), 2.0
    it, reason = 0, "max_iterations"
    history = [cost]
    accepted_at = [0]          # instrumentation only: the iteration every entry of `history` was accepted at
    accepted_s = [time.perf_counter() - t0]   # ... and the time since the solve began (instrumentation only)
    while it < max_iterations:
        it += 1
        if np.abs(gf).max() <= gradient_tolerance:
            reason = "gradient_tolerance"
            break
        d2 = np.clip(band[u], 1e-6, 1e32)
        A = band.copy()
        A[u] += d2 / radius
        try:
            step = -solveh_banded(A, gf, lower=False, check_finite=False)
        except np.linalg.LinAlgError:
            radius /= decrease
            decrease *= 2
            continue
        xf = x.ravel()[perm]
        if np.linalg.norm(step) <= parameter_tolerance * (np.linalg.norm(xf) + parameter_tolerance):
            reason = "parameter_tolerance"
            break
        cand = x.copy().ravel()
        cand[perm] += step
        cand = cand.reshape(-1, 4)
        cand[:, 3] = normalize_angle(cand[:, 3])
        new_cost, new_gf, (new_band, new_V) = problem.evaluate_reduced(cand)
        model_decrease = -(gf @ step + 0.5 * step @ problem.reduced_matvec(V, step))
        rho = (cost - new_cost) / model_decrease if model_decrease > 0 else -1.0
        if verbose:
            print(f"  it {it}: cost {cost:.6e} -> {new_cost:.6e} rho {rho:.3f} radius {radius:.2e} "
                  f"|step| {np.linalg.norm(step):.3e}")
        if rho > 1e-3:
            rel = abs(cost - new_cost) / max(cost, 1e-300)
            x, cost, gf, band, V = cand, new_cost, new_gf, new_band, new_V
            history.append(cost)
            accepted_at.append(it)
            accepted_s.append(time.perf_counter() - t0)
            radius = min(radius / max(1.0 / 3.0, 1.0 - (2.0 * rho - 1.0) ** 3), 1e16)
            decrease = 2.0
            if rel <= function_tolerance:
                reason = "function_tolerance"
                break
        else:
            radius /= decrease
            decrease *= 2.0
        if time.perf_counter() - t0 > max_seconds:
            reason = "max_solver_time"
            break
    return x, {"iterations": it, "evaluations": problem.evaluations, "termination": reason,
               "final_cost": cost, "initial_cost": history[0],
               "cost_history": list(zip(accepted_at, history)),
               "seconds_history": accepted_s,
               "seconds": time.perf_counter() - t0,
               # instrumentation only: what the registration backend took -- kernels,
               # copy of the fused buffer, all-reduce -- and what this harness' own assembly + banded
               # Cholesky took (Ceres does that part in the real system)
               "backend_seconds": problem.backend_seconds,
               "host_linear_algebra_seconds": time.perf_counter() - t0 - problem.backend_seconds}


def zero_registration_backend(n_nodes, n_constraints):
    """Backend for `exclude_registration_constraints = true` (pose_graph.cpp:74-83)."""
    buf = np.zeros(1 + 20 * n_nodes + 16 * n_constraints)
    return lambda poses: buf


def optimize_two_stage(backend, n_nodes, pairs, edges, poses0, new_loop_closures, **solve_kw):
    """PoseGraphInterface::optimize (pose_graph_interface.cpp:177-198): when new loop
    closures were added, first optimise WITHOUT the registration constraints (odometry +
    loop closures only, :182-188), then run the full problem from that result (:191)."""
    poses = np.array(poses0, np.float64)
    summaries = []
    if new_loop_closures:
        pre = Problem(zero_registration_backend(n_nodes, len(pairs)), n_nodes, pairs, edges)
        poses, s = solve(pre, poses, **solve_kw)
        summaries.append(s)
    full = Problem(backend, n_nodes, pairs, edges)
    poses, s = solve(full, poses, **solve_kw)
    summaries.append(s)
    return poses, summaries
