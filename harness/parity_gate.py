"""Same-run parity gate of bench.py: a bounded sample of the constraints the TIMED
launches evaluated is re-evaluated by the checker and compared

  * row by row -- the materialised f32 rows the launch left in its output buffers against the
    reference's own RegistrationCostFunction::Evaluate (oracle/_ref: registration_cost_function.cpp
    compiled from /root/reference, when the prebuilt library travelled with the snapshot; the port
    oracle/reg_oracle.c, which tests/test_ref_pin.py pins to it value for value, otherwise): the GPU's
    f32 value must be the f32 rounding of the reference's f64 value, EXACTLY
    (registration_cost_function.cpp:161-170, 254-291);
  * block by block -- the fused pass's 45 numbers per constraint [cost | J^T r | upper J^T J] against
    oracle reg_evaluate_normal, 1e-6 relative to the block's largest entry.

CHECKER SIDE: imports oracle/ (tests and bench only, never the product)."""
import os
from concurrent.futures import ThreadPoolExecutor

import numpy as np


def choose(n_rows, live, n_total=8, n_partial=3, n_dead=1):
    """constraint indices to check: some whose tiles are partly culled, one fully culled (if there is
    one), the rest the heaviest fully live ones; deterministic"""
    n_rows, live = np.asarray(n_rows, np.int64), np.asarray(live, np.int64)
    idx = np.arange(len(n_rows))
    dead = idx[(live == 0) & (n_rows > 0)]
    partial = idx[(live > 0) & (live < 0.9 * n_rows)]
    full = idx[live >= 0.9 * n_rows]
    out = list(dead[:n_dead])
    # partly culled: spread over the culled fraction
    if len(partial):
        frac = live[partial] / np.maximum(n_rows[partial], 1)
        order = partial[np.argsort(frac, kind="stable")]
        take = np.unique(np.linspace(0, len(order) - 1, min(n_partial, len(order))).round().astype(int))
        out += [int(order[t]) for t in take]
    rest = [int(c) for c in full[np.argsort(-n_rows[full], kind="stable")] if c not in out]
    more = [int(c) for c in partial if c not in out]
    for c in rest + more + [int(c) for c in dead if c not in out]:
        if len(out) >= n_total:
            break
        out.append(c)
    return out[:n_total]


def check(capi, ctx, torch, label, layers_of, dev_submaps, batch, pairs_local, poses, chosen, rows, voxel_size,
          vps=16, threads=None, submap_of_node=lambda n: n):
    """layers_of(k) -> (block_index, tsdf_d, tsdf_w, esdf_d, esdf_obs) host arrays of submap k;
    dev_submaps[k]: the device submap whose kVoxels points the launch read; rows = (residuals,
    jac_ref, jac_read) torch buffers the timed launch wrote, or None to evaluate the chosen
    constraints' rows now (same batch, same poses).  Returns the `parity` entry for this workload."""
    from oracle import pyoracle as orc
    try:
        from oracle import ref_reg
        use_ref = ref_reg.available()
    except Exception:
        use_ref = False
    ro = batch.row_offsets()
    if rows is None:
        R = int(ro[-1])
        rows = (torch.empty(R, dtype=torch.float32, device="cuda"), torch.empty((R, 4), dtype=torch.float32, device="cuda"),
                torch.empty((R, 4), dtype=torch.float32, device="cuda"))
        batch.evaluate_points(poses, rows[0].data_ptr(), rows[1].data_ptr(), rows[2].data_ptr())
        ctx.synchronize()
    torch.cuda.synchronize()
    _, normal = batch.evaluate_normal(poses)
    sub = lambda n: int(submap_of_node(int(n)))      # pose-graph node -> the submap it carries a pose of
    need = sorted({sub(s) for c in chosen for s in pairs_local[c]})
    host = {k: layers_of(k) for k in need}
    pts = {}
    for c in chosen:
        a = sub(pairs_local[c][0])
        if a not in pts:
            pts[a] = dev_submaps[a].download_points(capi.POINTS_VOXELS)
    orc_layers = {k: orc.Layer(voxel_size, vps, host[k][0], host[k][3], host[k][4]) for k in need}
    ref_sub = {}
    if use_ref:
        for k in need:
            bi, td, tw, ed, eo = host[k]
            ref_sub[k] = ref_reg.Submap(k, np.zeros(4), voxel_size, vps, bi, td, tw, ed, eo)

    def one(c):
        na, nb = int(pairs_local[c][0]), int(pairs_local[c][1])
        a, b = sub(na), sub(nb)
        xyz, d, w = pts[a]
        s = slice(int(ro[c]), int(ro[c + 1]))
        g = [t[s].cpu().numpy() for t in rows]
        if use_ref:
            # the reference submap's sampler holds the device's point list in the device's order
            ref_sub[a].set_points(ref_reg.POINTS_VOXELS, xyz, d, w)
            cf = ref_reg.RegistrationCostFunction(ref_sub[a], ref_sub[b])
            ok, r0, j0, j1 = cf.Evaluate(poses[na], poses[nb])
        else:
            ok, r0, j0, j1 = orc.reg_evaluate(orc_layers[b], xyz, d, w, poses[na], poses[nb])
        want = [r0, j0, j1]
        n_bad, max_rel = 0, 0.0
        for gg, ww in zip(g, want):
            w32 = ww.astype(np.float32)
            bad = gg != w32
            n_bad += int(bad.sum())
            if bad.any():
                scale = np.maximum(np.abs(ww), 1e-3 * np.abs(ww).max())
                max_rel = max(max_rel, float((np.abs(gg.astype(np.float64) - ww) / np.maximum(scale, 1e-300))[bad].max()))
        ok2, cost, jtr, jtj = orc.reg_evaluate_normal(orc_layers[b], xyz, d, w, poses[na], poses[nb])
        blk = np.r_[cost, jtr, np.asarray(jtj).reshape(-1)]
        got = normal[c]
        if len(blk) != len(got):                       # oracle returns the 36 upper-triangle entries
            raise RuntimeError(f"normal block sizes differ: {len(blk)} vs {len(got)}")
        # per part, relative to the part's largest entry (cost | J^T r | J^T J)
        rel = 0.0
        for lo, hi in ((0, 1), (1, 9), (9, 45)):
            ref_max = np.abs(blk[lo:hi]).max()
            if ref_max > 0:
                rel = max(rel, float(np.abs(got[lo:hi] - blk[lo:hi]).max() / ref_max))
            elif np.abs(got[lo:hi]).max() != 0:
                rel = float("inf")
        with_corr = int((np.abs(j0).sum(1) > 0).sum())
        return dict(rows=int(s.stop - s.start), mismatched_values=n_bad, max_rel=max_rel, ok=bool(ok), normal_rel=rel,
                    with_correspondence=with_corr)

    threads = threads or min(len(chosen), os.cpu_count() or 1, 16)
    # holders mutate a shared ref submap's sampler: constraints that share a reference submap run in turn
    groups = {}
    for c in chosen:
        groups.setdefault(sub(pairs_local[c][0]), []).append(c)

    def run_group(cs):
        return [(c, one(c)) for c in cs]
    with ThreadPoolExecutor(max(threads, 1)) as ex:
        res = dict(kv for part in ex.map(run_group, groups.values()) for kv in part)
    rows_checked = sum(r["rows"] for r in res.values())
    values = 9 * rows_checked
    bad = sum(r["mismatched_values"] for r in res.values())
    return {"workload": label, "constraints_checked": len(chosen),
            "rows_checked": rows_checked, "values_checked": values,
            "rows_with_correspondence": sum(r["with_correspondence"] for r in res.values()),
            "fully_culled_constraints": sum(1 for r in res.values() if r["with_correspondence"] == 0),
            "mismatched_values": bad, "exact": bad == 0,
            "max_rel": max(r["max_rel"] for r in res.values()),
            "fused_blocks_max_rel": max(r["normal_rel"] for r in res.values()),
            "fused_blocks_within_1e-6": all(r["normal_rel"] <= 1e-6 for r in res.values()),
            "checker": "oracle/_ref (the reference's registration_cost_function.cpp)" if use_ref else
                       "oracle/reg_oracle.c (port; pinned to the reference source by tests/test_ref_pin.py)",
            "rule": "rows: GPU f32 == f32(checker f64), every value; fused 45-blocks: <= 1e-6 of the part's largest entry"}


def merge(entries):
    """the line's `parity` object"""
    entries = [e for e in entries if e]
    if not entries:
        return None
    return {"checked": sum(e["values_checked"] for e in entries),
            "constraints_checked": sum(e["constraints_checked"] for e in entries),
            "max_rel": max(e["max_rel"] for e in entries),
            "exact": all(e["exact"] for e in entries),
            "fused_blocks_max_rel": max(e["fused_blocks_max_rel"] for e in entries),
            "fused_blocks_within_1e-6": all(e["fused_blocks_within_1e-6"] for e in entries),
            "checker": entries[0]["checker"], "rule": entries[0]["rule"],
            "per_workload": entries}
