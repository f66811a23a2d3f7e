#!/usr/bin/env python3
"""Turns the rocprofv3 CSVs collected by profiles/collect.sh into the committed
summaries: profiles/rNN_kernel_stats.json, profiles/rNN_pmc_hbm.json and
profiles/hbm_traffic.json (read by bench.py for roofline.traffic).

HBM bytes follow MI355X_MICROARCH.md "HBM": FETCH_SIZE / WRITE_SIZE are in KiB and
are uncalibrated on gfx950 (FETCH_SIZE reads 1/2 of a wide coalesced stream), so
each counter is corrected by the factor measured on a calibration launch of the
same kernel with a known byte count (every evaluation without correspondence:
exactly 20 B read, 36 B written per residual)."""
import argparse
import csv
import glob
import json
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
KERNEL = "reg_eval_points_kernel"


def rows(pattern):
    out = []
    for f in sorted(glob.glob(os.path.join(ROOT, "gpurun_out", pattern), recursive=True)):
        with open(f, newline="") as fh:
            out += list(csv.DictReader(fh))
    return out


def counter(prefix, name):
    r = [x for x in rows(f"{prefix}/**/*counter_collection.csv")
         if KERNEL in x.get("Kernel_Name", "") and x.get("Counter_Name") == name]
    r.sort(key=lambda x: int(x.get("Dispatch_Id", 0)))
    return [float(x["Counter_Value"]) for x in r]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--round", default="01")
    a = ap.parse_args()
    tag = f"r{int(a.round):02d}"
    # ---- kernel stats of the bench command -----------------------------------
    stats = rows("prof_stats/**/*kernel_stats.csv")
    bench = None
    bpath = os.path.join(ROOT, "gpurun_out", "prof_stats_bench.json")
    if os.path.exists(bpath):
        for line in open(bpath):
            if line.startswith("{"):
                bench = json.loads(line)
    summ = {"command": "rocprofv3 --kernel-trace --stats -f csv -- python bench.py",
            "kernels": [{k: v for k, v in x.items()} for x in stats[:12]],
            "bench_line": bench}
    for x in stats:
        if KERNEL in x.get("Name", ""):
            avg_ns = float(x.get("AverageNs", x.get("Average", 0)))
            summ["dominant_kernel"] = {"name": x["Name"], "calls": int(x.get("Calls", 0)),
                                       "avg_ms_rocprof": avg_ns / 1e6}
            if bench:
                summ["dominant_kernel"]["avg_ms_bench_hip_events"] = bench["roofline"]["kernel_ms"]
    json.dump(summ, open(os.path.join(ROOT, "profiles", f"{tag}_kernel_stats.json"), "w"), indent=1)
    # ---- HBM traffic ------------------------------------------------------------
    fetch, write = counter("prof_fetch", "FETCH_SIZE"), counter("prof_write", "WRITE_SIZE")
    pmc_bench = None
    ppath = os.path.join(ROOT, "gpurun_out", "prof_fetch_bench.json")
    if os.path.exists(ppath):
        for line in open(ppath):
            if line.startswith("{"):
                pmc_bench = json.loads(line)
    res = {"fetch_KiB_per_dispatch": fetch, "write_KiB_per_dispatch": write}
    if fetch and write and pmc_bench and len(fetch) > 1 and len(write) > 1:
        R = pmc_bench["roofline"]["units_per_launch"]
        known_r, known_w = 20.0 * R, 36.0 * R
        cf = known_r / (fetch[0] * 1024.0)        # dispatch 0 = calibration launch
        cw = known_w / (write[0] * 1024.0)
        fr = sum(fetch[1:]) / len(fetch[1:]) * 1024.0 * cf
        wr = sum(write[1:]) / len(write[1:]) * 1024.0 * cw
        res.update({"residuals_per_launch": R, "calibration": {
            "known_read_bytes": known_r, "known_write_bytes": known_w,
            "fetch_correction": cf, "write_correction": cw},
            "hbm_read_bytes_per_launch": fr, "hbm_write_bytes_per_launch": wr,
            "hbm_bytes_per_launch": fr + wr,
            "algorithmic_bytes_88": 88.0 * R,
            "n_gpus": 1})
        json.dump({"residuals_per_launch": R, "n_gpus": 1, "hbm_bytes_per_launch": fr + wr,
                   "source": f"profiles/{tag}_pmc_hbm.json"},
                  open(os.path.join(ROOT, "profiles", "hbm_traffic.json"), "w"), indent=1)
    json.dump(res, open(os.path.join(ROOT, "profiles", f"{tag}_pmc_hbm.json"), "w"), indent=1)
    print(json.dumps({"stats": summ.get("dominant_kernel"), "pmc": {k: v for k, v in res.items()
                                                                    if not k.endswith("_dispatch")}}, indent=1))


if __name__ == "__main__":
    main()
