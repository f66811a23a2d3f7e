#!/usr/bin/env python3
"""Turns the rocprofv3 CSVs collected by profiles/collect.sh into the committed summaries:
profiles/rNN_kernel_stats.json (per kernel AND per workload: dispatches are grouped by grid size),
profiles/rNN_pmc_hbm.json, profiles/rNN_pmc_fused.json, profiles/rNN_sq_breakdown.json and
profiles/hbm_traffic.json (read by bench.py for roofline.traffic), and copies the raw CSVs.

HBM bytes, following MI355X_MICROARCH.md "HBM":
  reads  : FETCH_SIZE is RDREQ x 64 B on gfx950, i.e. half of a stream of 128-B requests.  Instead
           of doubling it blindly the read requests are counted per size class
           (TCC_EA0_RDREQ_{32B,64B,128B}_sum): bytes = 32 n32 + 64 n64 + 128 n128, exact for any
           request mix; FETCH_SIZE of a second pass is reported beside it (x2 must agree).
  writes : WRITE_SIZE is uncalibrated, so it is scaled by (known bytes / counted bytes) of the
           calibration dispatches of the same kernel, in which every residual writes exactly 36 B
           (outputs are written once, so the known figure is exact; measured factor ~0.95 with
           the kernel's non-temporal stores).  The fused kernel uses plain stores (x0.998).
Reads are NOT calibrated against "20 B per point": constraints that share a reference submap
re-read its points out of L2 / Infinity Cache, so the true fabric read volume is below that."""
import argparse
import collections
import csv
import glob
import json
import os
import shutil

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
KERNEL, FUSED, TSDF = "reg_eval_points_kernel", "reg_eval_reduce", "tsdf_integrate"   # (tsdf_integrate_coop_kernel since round 5)


def rows(pattern):
    """csv rows of every file matching the pattern under gpurun_out/ (collect.sh gzips the large traces: gpurun brings at
    most 64 MiB back)"""
    import gzip
    out = []
    for f in sorted(glob.glob(os.path.join(ROOT, "gpurun_out", pattern), recursive=True) +
                    glob.glob(os.path.join(ROOT, "gpurun_out", pattern + ".gz"), recursive=True)):
        with (gzip.open(f, "rt", newline="") if f.endswith(".gz") else open(f, newline="")) as fh:
            out += list(csv.DictReader(fh))
    return out


def is_cost_only(name):
    """the cost-only instantiation of the fused tile kernel (round 6): reg_eval_reduce_lean_kernel<..., true>"""
    import re
    return bool(re.search(r"reg_eval_reduce_lean_kernel<[^>]*true>", name))


def is_f64_rows(name):
    """the OUT = double instantiation of the materialising kernel (vgx_reg_batch_evaluate_points_f64 and the drop-in
    vgx_reg_evaluate): kept apart from the f32 headline kernel, whose name and grids it shares"""
    import re
    return bool(re.search(r"reg_eval_points(_single)?_kernel<\d+, \d+, double", name))


def dispatches(prefix, kernel, cost_only=False):
    """[{counter: value, "grid": n}] per dispatch of `kernel`, in dispatch order (cost_only: that instantiation alone,
    else every other one)"""
    d = {}
    for x in rows(f"{prefix}/**/*counter_collection.csv"):
        if kernel in x.get("Kernel_Name", "") and is_cost_only(x.get("Kernel_Name", "")) == cost_only and not is_f64_rows(x.get("Kernel_Name", "")):
            e = d.setdefault(int(x["Dispatch_Id"]), {"grid": int(x["Grid_Size"])})
            e[x["Counter_Name"]] = float(x["Counter_Value"])
    return [d[k] for k in sorted(d)]


def read_bytes(c):
    return (32.0 * c.get("TCC_EA0_RDREQ_32B_sum", 0) + 64.0 * c.get("TCC_EA0_RDREQ_64B_sum", 0) +
            128.0 * c.get("TCC_EA0_RDREQ_128B_sum", 0))


def bench_line(name):
    """the FULL result object of a bench.py run: its --detail file (round 4 on: stdout carries only the compact
    summary line), else the stdout line of the earlier rounds"""
    pd = os.path.join(ROOT, "gpurun_out", name.replace("_bench.json", "_detail.json"))
    if name.endswith("_bench.json") and os.path.exists(pd):
        return json.load(open(pd))
    p = os.path.join(ROOT, "gpurun_out", name)
    if os.path.exists(p):
        for line in open(p):
            if line.startswith("{"):
                return json.loads(line)
    return None


def mean(v):
    v = list(v)
    return sum(v) / len(v) if v else None


def by_grid(trace, kernel, cost_only=False, f64_rows=False):
    """kernel-trace rows of one kernel grouped per grid size, in order of first appearance:
    [(grid, [duration_ns, ...])]"""
    g = collections.OrderedDict()
    for x in trace:
        if (kernel in x.get("Kernel_Name", "") and is_cost_only(x.get("Kernel_Name", "")) == cost_only
                and is_f64_rows(x.get("Kernel_Name", "")) == f64_rows):
            g.setdefault(int(x["Grid_Size_X"]), []).append(int(x["End_Timestamp"]) - int(x["Start_Timestamp"]))
    return list(g.items())


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--round", default="04")
    a = ap.parse_args()
    tag = f"r{int(a.round):02d}"
    N_CAL = 2                                            # bench.py --calibrate: two far-pose launches
    # ---- kernel stats of the bench command, per workload -------------------------------------
    stats = rows("prof_stats/**/*kernel_stats.csv")
    trace = rows("prof_stats/**/*kernel_trace.csv")
    bench = bench_line("prof_stats_bench.json")
    summ = {"command": "rocprofv3 --kernel-trace --stats -f csv -- python bench.py --gpus 1 --steps 20 --warmup 5 (25 passes per step)",
            "kernels": [{k: v for k, v in x.items()} for x in stats[:14]],
            "bench_line": bench, "per_workload": {}}
    pw = summ["per_workload"]

    def entry(durs, hip_ms=None, **extra):
        e = {"calls": len(durs), "avg_ms_rocprof": mean(durs) / 1e6, "min_ms": min(durs) / 1e6, "max_ms": max(durs) / 1e6}
        if hip_ms is not None:
            e["avg_ms_bench_hip_events"] = hip_ms
            e["rocprof_over_hip_events"] = e["avg_ms_rocprof"] / hip_ms
        e.update(extra)
        return e
    pts = by_grid(trace, KERNEL)
    if bench and len(pts) >= 1:
        # bench.py's placement trials (vgx_reg_batch_choose_outputs: one warm-up + 3 launches per trial, on candidate arrays
        # of which about half are slow ones) come first on this grid: kept apart from the launches the line's time is about
        n_skip = 0
        pl = (bench.get("roofline") or {}).get("output_placement") or {}      # (bench = the run's --detail object)
        n_skip = 4 * sum(1 for t in (pl.get("ms_trials") or []) if t and t > 0)
        durs3 = pts[0][1]
        if 0 < n_skip < len(durs3):
            pw["config3_points_placement_trials"] = entry(durs3[:n_skip], grid=pts[0][0],
                                                          what="launches of vgx_reg_batch_choose_outputs, before the timed region")
            durs3 = durs3[n_skip:]
        pw["config3_points"] = entry(durs3, bench["roofline"]["kernel_ms"], grid=pts[0][0])
    if bench and len(pts) >= 2 and bench.get("roofline_full_overlap"):
        fo_b = bench["roofline_full_overlap"]
        durs = pts[1][1]
        plain = (fo_b.get("plain_order") or {}).get("kernel_ms")
        if plain and len(durs) % 2 == 0:
            # bench.py runs the workload twice with the same grid: shipped launch order first, then plain
            half = len(durs) // 2
            pw["full_overlap_points_plain_order"] = entry(durs[half:], plain, grid=pts[1][0])
            durs = durs[:half]
        pw["full_overlap_points"] = entry(durs, fo_b["kernel_ms"], grid=pts[1][0])
    # the f64-rows instantiation (vgx_reg_batch_evaluate_points_f64) on config 3's grid: two warm-up launches, then the timed ten
    for grid, durs in by_grid(trace, "reg_eval_points_kernel<", f64_rows=True)[:1]:
        f64_ms = ((bench or {}).get("roofline") or {}).get("f64_rows_ms")
        pw["config3_points_f64_rows"] = entry(durs[2:] if len(durs) > 2 else durs, f64_ms, grid=grid)
    red = by_grid(trace, FUSED)
    # grids in order of first appearance in bench.py: config 3, full overlap, shipped (sampled), the two
    # shards of the in-process multi-context section, config 5
    names = ["config3_fused", "full_overlap_fused", "shipped_config_fused", "multi_context_shard0_fused",
             "multi_context_shard1_fused", "config5_fused"]
    for (grid, durs), nm in zip(red, names):
        if nm == "shipped_config_fused" and bench and (bench.get("shipped_config") or {}).get("apron_bricks") and len(durs) % 2 == 0:
            # bench.py evaluates the shipped configuration twice on the same grid: the default first (quad bricks made
            # on demand), apron bricks second
            half = len(durs) // 2
            pw["shipped_config_apron_bricks_fused"] = entry(durs[half:], grid=grid)
            durs = durs[:half]
        pw[nm] = entry(durs, grid=grid)
    if bench and "config3_fused" in pw:
        pw["config3_fused"]["bench_stream_ms_per_step_incl_finalize_assemble"] = bench["fused"]["stream_ms_per_step"]
    # the cost-only instantiation (vgx_reg_batch_evaluate_cost), same grids in the same order
    for (grid, durs), nm in zip(by_grid(trace, FUSED, cost_only=True), ["config3_fused_cost_only", "full_overlap_fused_cost_only"]):
        pw[nm] = entry(durs, grid=grid)
    # the bench's own TSDF section only: the config-2 session that follows integrates scans of the
    # same size (synth_city_scan_kernel marks where it starts)
    cut = next((i for i, x in enumerate(trace) if "synth_city_scan" in x.get("Kernel_Name", "")), len(trace))
    if bench and bench.get("tsdf"):
        for name, v in bench["tsdf"].items():
            grid = (v["points_per_scan"] + 255) // 256 * 256
            t = [(int(x["Start_Timestamp"]), int(x["End_Timestamp"])) for x in trace[:cut]
                 if TSDF in x.get("Kernel_Name", "") and int(x["Grid_Size_X"]) == grid]
            n_timed = v["scans_timed"]
            if len(t) < 1 + n_timed:
                continue
            # tsdf_bench's first pass over this sensor: one warm-up scan, then the timed back-to-back scans
            timed = t[1:1 + n_timed]
            ksum = sum(e - s_ for s_, e in timed)
            span = timed[-1][1] - timed[0][0]
            pw["tsdf_" + name] = {
                "grid": grid, "calls": n_timed, "what": "the back-to-back scans bench.py times (first pass, after the warm-up scan)",
                "avg_kernel_ms_rocprof": ksum / n_timed / 1e6,
                "span_ms_per_scan_rocprof": span / n_timed / 1e6,
                "scan_wall_over_kernel_time": span / ksum,
                "gaps_us": [round((timed[i + 1][0] - timed[i][1]) / 1e3, 1) for i in range(n_timed - 1)],
                "back_to_back_ms_per_scan_bench_hip_events": v["ms_per_scan"],
                "isolated_kernel_ms_bench_hip_events": v["roofline"]["kernel_ms"]}
    json.dump(summ, open(os.path.join(ROOT, "profiles", f"{tag}_kernel_stats.json"), "w"), indent=1)
    # ---- HBM traffic of the dominant kernel, both workloads ---------------------------------------
    rd, fetch, write = dispatches("prof_rd", KERNEL), dispatches("prof_fetch", KERNEL), dispatches("prof_write", KERNEL)
    pmc_bench = bench_line("prof_rd_bench.json")
    res = {"calibration_dispatches": N_CAL}
    traffic = {}
    if pmc_bench and len(rd) > N_CAL and len(write) > N_CAL:
        main_grid = rd[0]["grid"]
        R = pmc_bench["roofline"]["units_per_launch"]
        known_w = 36.0 * R
        cw = known_w / (mean(c["WRITE_SIZE"] for c in write[:N_CAL]) * 1024.0)
        res["calibration"] = {"known_write_bytes": known_w, "write_correction": cw,
                              "read_bytes_calibration_launch": mean(read_bytes(c) for c in rd[:N_CAL]),
                              "read_bytes_note": "every tile of the calibration launch is culled: no points read "
                                                 "(tile descriptors + flags only; 20 B per point would be %.3g)" % (20.0 * R)}
        groups = {"config3": (lambda c: c["grid"] == main_grid, R, pmc_bench["roofline"]["kernel_ms"])}
        fo = pmc_bench.get("roofline_full_overlap")
        if fo:
            groups["full_overlap"] = (lambda c: c["grid"] != main_grid, fo["units_per_launch"], fo["kernel_ms"])
        for name, (sel, units, hip_ms) in groups.items():
            r_ = [c for c in rd[N_CAL:] if sel(c)]
            w_ = [c for c in write[N_CAL:] if sel(c)]
            f_ = [c for c in fetch[N_CAL:] if sel(c)]
            if not r_ or not w_:
                continue
            fr = mean(read_bytes(c) for c in r_)
            wr = mean(c["WRITE_SIZE"] for c in w_) * 1024.0 * cw
            e = {"dispatches": len(r_), "residuals_per_launch": units, "hbm_read_bytes_per_launch": fr,
                 "hbm_write_bytes_per_launch": wr, "hbm_bytes_per_launch": fr + wr,
                 "algorithmic_bytes_88": 88.0 * units, "traffic_over_algorithmic_88": (fr + wr) / (88.0 * units),
                 "kernel_ms_under_pmc_hip_events": hip_ms,
                 "hbm_GBs_at_pmc_run_kernel_ms": (fr + wr) / (hip_ms * 1e-3) / 1e9}
            if f_:
                e["fetch_size_x2_over_exact"] = mean(c["FETCH_SIZE"] for c in f_) * 1024.0 * 2.0 / fr
            key = name + "_points"
            if key in pw:
                e["hbm_GBs_at_rocprof_avg"] = (fr + wr) / (pw[key]["avg_ms_rocprof"] * 1e-3) / 1e9
            res[name] = e
            traffic[name] = e
        # the full-overlap workload in plain launch order (VGX_POINTS_TILE_ORDER=0 passes)
        rdp, wrp = dispatches("prof_rd_plain", KERNEL), dispatches("prof_write_plain", KERNEL)
        pb = bench_line("prof_rd_plain_bench.json")
        if fo and pb and len(rdp) > N_CAL and len(wrp) > N_CAL:
            r_ = [c for c in rdp[N_CAL:] if c["grid"] != main_grid]
            w_ = [c for c in wrp[N_CAL:] if c["grid"] != main_grid]
            if r_ and w_:
                fr = mean(read_bytes(c) for c in r_)
                wr = mean(c["WRITE_SIZE"] for c in w_) * 1024.0 * cw
                hip_ms = pb["roofline_full_overlap"]["kernel_ms"]
                e = {"dispatches": len(r_), "residuals_per_launch": fo["units_per_launch"],
                     "hbm_read_bytes_per_launch": fr, "hbm_write_bytes_per_launch": wr, "hbm_bytes_per_launch": fr + wr,
                     "algorithmic_bytes_88": 88.0 * fo["units_per_launch"],
                     "traffic_over_algorithmic_88": (fr + wr) / (88.0 * fo["units_per_launch"]),
                     "kernel_ms_under_pmc_hip_events": hip_ms,
                     "hbm_GBs_at_pmc_run_kernel_ms": (fr + wr) / (hip_ms * 1e-3) / 1e9}
                if "full_overlap_points_plain_order" in pw:
                    e["hbm_GBs_at_rocprof_avg"] = (fr + wr) / (pw["full_overlap_points_plain_order"]["avg_ms_rocprof"] * 1e-3) / 1e9
                res["full_overlap_plain_order"] = e
                traffic["full_overlap_plain_order"] = e
        if "config3" in traffic:
            t = {"residuals_per_launch": traffic["config3"]["residuals_per_launch"], "n_gpus": 1,
                 "hbm_bytes_per_launch": traffic["config3"]["hbm_bytes_per_launch"], "source": f"profiles/{tag}_pmc_hbm.json"}
            if "full_overlap" in traffic:
                t["full_overlap"] = {"residuals_per_launch": traffic["full_overlap"]["residuals_per_launch"],
                                     "hbm_bytes_per_launch": traffic["full_overlap"]["hbm_bytes_per_launch"]}
            if "full_overlap_plain_order" in traffic:
                t["full_overlap_plain_order"] = {
                    "residuals_per_launch": traffic["full_overlap_plain_order"]["residuals_per_launch"],
                    "hbm_bytes_per_launch": traffic["full_overlap_plain_order"]["hbm_bytes_per_launch"]}
            json.dump(t, open(os.path.join(ROOT, "profiles", "hbm_traffic.json"), "w"), indent=1)
    json.dump(res, open(os.path.join(ROOT, "profiles", f"{tag}_pmc_hbm.json"), "w"), indent=1)
    # ---- fused kernel, both workloads --------------------------------------------------------------
    frd, fwr = dispatches("prof_rd", FUSED), dispatches("prof_write", FUSED)
    fused = {}
    if frd and fwr and pmc_bench:
        g0 = frd[0]["grid"]
        grids = []
        for c in frd:
            if c["grid"] not in grids:
                grids.append(c["grid"])
        g1 = grids[1] if len(grids) > 1 else None
        g2 = grids[2] if len(grids) > 2 else None
        sh = pmc_bench.get("shipped_config")
        sh_fb = None
        if sh:      # the shipped (sampled) configuration: priced per residual, 52 B each (scattered draws)
            sh_fb = {"algorithmic_bytes_per_step": 52.0 * sh["residuals_per_evaluation"],
                     "evaluations": sh["residuals_per_evaluation"], "with_correspondence": None,
                     "loaded_after_culling": sh["residuals_per_evaluation"]}
        has_quad = bool(sh and sh.get("apron_bricks"))          # two runs on the same grid: default (quad), then apron
        for name, sel, fb in (("config3", lambda c: c["grid"] == g0, pmc_bench.get("fused")),
                              ("full_overlap", lambda c: g1 is not None and c["grid"] == g1,
                               (pmc_bench.get("roofline_full_overlap") or {}).get("fused")),
                              ("shipped", lambda c: g2 is not None and c["grid"] == g2, sh_fb),
                              ("shipped_apron", lambda c: g2 is not None and c["grid"] == g2, sh_fb if has_quad else None)):
            r_ = [c for c in frd if sel(c)]
            w_ = [c for c in fwr if sel(c)]
            if name.startswith("shipped") and has_quad and len(r_) % 2 == 0 and len(w_) % 2 == 0:
                # the default (quad bricks on demand) first, apron bricks second (same grid)
                hr, hw = len(r_) // 2, len(w_) // 2
                r_, w_ = (r_[:hr], w_[:hw]) if name == "shipped" else (r_[hr:], w_[hw:])
            elif name == "shipped_apron":
                continue
            if not r_ or not w_ or not fb:
                continue
            e = {"kernel": "reg_eval_reduce_lean_kernel", "dispatches": len(r_),
                 "hbm_read_bytes_per_launch": mean(read_bytes(c) for c in r_),
                 "hbm_write_bytes_per_launch": mean(c["WRITE_SIZE"] for c in w_) * 1024.0 * 0.998,
                 "algorithmic_bytes_per_step": fb["algorithmic_bytes_per_step"],
                 "evaluations": fb["evaluations"], "with_correspondence": fb["with_correspondence"],
                 "loaded_after_culling": fb["loaded_after_culling"]}
            e["hbm_bytes_per_launch"] = e["hbm_read_bytes_per_launch"] + e["hbm_write_bytes_per_launch"]
            e["traffic_over_algorithmic"] = e["hbm_bytes_per_launch"] / e["algorithmic_bytes_per_step"]
            key = {"shipped": "shipped_config_fused", "shipped_apron": "shipped_config_apron_bricks_fused"}.get(name, name + "_fused")
            if key in pw:
                e["avg_ms_rocprof"] = pw[key]["avg_ms_rocprof"]
                e["hbm_GBs_at_rocprof_avg"] = e["hbm_bytes_per_launch"] / (e["avg_ms_rocprof"] * 1e-3) / 1e9
                e["hbm_time_ms_at_8TBs"] = e["hbm_bytes_per_launch"] / 8e12 * 1e3
                e["frac_of_hbm_time"] = e["hbm_time_ms_at_8TBs"] / e["avg_ms_rocprof"]
            fused[name] = e
        json.dump(fused, open(os.path.join(ROOT, "profiles", f"{tag}_pmc_fused.json"), "w"), indent=1)
        # what bench.py replays as fused.*.traffic_from_profiles / hbm_frac
        tpath = os.path.join(ROOT, "profiles", "hbm_traffic.json")
        if os.path.exists(tpath):
            t = json.load(open(tpath))
            t["fused"] = {k: {"evaluations": v["evaluations"], "hbm_bytes_per_launch": v["hbm_bytes_per_launch"],
                              "avg_ms_rocprof": v.get("avg_ms_rocprof"), "source": f"profiles/{tag}_pmc_fused.json"}
                          for k, v in fused.items()}
            json.dump(t, open(tpath, "w"), indent=1)
    # ---- SQ wave-cycle breakdown ---------------------------------------------------------------------
    d = collections.OrderedDict()
    for r in rows("prof_sq/**/*counter_collection.csv"):
        kn = r["Kernel_Name"]
        k = ("fused" if "reduce" in kn else "tsdf_integrate" if "tsdf_integrate" in kn else
             "tsdf_reproducible_" + ("apply" if "det_apply" in kn else "seen") if "det_" in kn else "materialising")
        # the reproducible TSDF mode's grids change from scan to scan: one entry per kernel
        grid = 0 if k.startswith("tsdf_reproducible") else int(r["Grid_Size"])
        d.setdefault((k, grid), collections.defaultdict(list))[r["Counter_Name"]].append(float(r["Counter_Value"]))
    sq = {}
    for (k, grid), c in d.items():
        m = {n: sum(v) / len(v) for n, v in c.items()}
        wc = m.get("SQ_WAVE_CYCLES", 0) or 1
        sq[f"{k}@grid{grid}" if grid else k] = {"dispatches": len(next(iter(c.values()))), **m,
                                 "frac_wait_any": m.get("SQ_WAIT_ANY", 0) / wc,
                                 "frac_wait_inst_any": m.get("SQ_WAIT_INST_ANY", 0) / wc,
                                 "frac_active_inst_any": m.get("SQ_ACTIVE_INST_ANY", 0) / wc,
                                 "frac_active_inst_valu": m.get("SQ_ACTIVE_INST_VALU", 0) / wc,
                                 "valu_insts_per_wave": m.get("SQ_INSTS_VALU", 0) / (m.get("SQ_WAVES", 0) or 1),
                                 "salu_insts_per_wave": m.get("SQ_INSTS_SALU", 0) / (m.get("SQ_WAVES", 0) or 1)}
    if sq:
        json.dump(sq, open(os.path.join(ROOT, "profiles", f"{tag}_sq_breakdown.json"), "w"), indent=1)
    # ---- copies of the raw evidence ---------------------------------------------------
    for src, dst in (("prof_stats/**/*kernel_stats.csv", f"{tag}_rocprofv3_kernel_stats.csv"),
                     ("prof_rd/**/*counter_collection.csv", f"{tag}_pmc_rdreq_counter_collection.csv"),
                     ("prof_fetch/**/*counter_collection.csv", f"{tag}_pmc_fetch_counter_collection.csv"),
                     ("prof_write/**/*counter_collection.csv", f"{tag}_pmc_write_counter_collection.csv"),
                     ("prof_rd_plain/**/*counter_collection.csv", f"{tag}_pmc_rdreq_plain_order_counter_collection.csv"),
                     ("prof_write_plain/**/*counter_collection.csv", f"{tag}_pmc_write_plain_order_counter_collection.csv"),
                     ("prof_stats_bench.json", f"{tag}_bench_under_rocprof.json"),
                     ("bench_full.json", f"{tag}_bench_full.json")):
        found = sorted(glob.glob(os.path.join(ROOT, "gpurun_out", src), recursive=True))
        if found:
            shutil.copyfile(found[0], os.path.join(ROOT, "profiles", dst))
    # the per-dispatch kernel trace is large: keep the REG / TSDF kernels' rows only
    keep_rows = [x for x in trace if any(k in x.get("Kernel_Name", "") for k in (KERNEL, FUSED, TSDF, "reg_finalize", "reg_assemble", "mt_generate"))]
    if keep_rows:
        with open(os.path.join(ROOT, "profiles", f"{tag}_rocprofv3_kernel_trace_reg_tsdf.csv"), "w", newline="") as fh:
            cols = ["Kernel_Name", "Dispatch_Id", "Start_Timestamp", "End_Timestamp", "Grid_Size_X", "VGPR_Count", "SGPR_Count", "LDS_Block_Size"]
            w = csv.DictWriter(fh, fieldnames=cols, extrasaction="ignore")
            w.writeheader()
            for x in keep_rows:
                x = dict(x)
                x["Kernel_Name"] = x["Kernel_Name"].split("(")[0][-60:]
                w.writerow(x)
    print(json.dumps({"per_workload": pw, "pmc": {k: v for k, v in res.items() if k in ("config3", "full_overlap", "full_overlap_plain_order", "calibration")},
                      "fused": fused}, indent=1))


if __name__ == "__main__":
    main()
