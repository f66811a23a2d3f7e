#!/usr/bin/env python3
"""Runs harness/chain_compare.py (GPU chain vs oracle-only chain on a cut of the config-2 session)
in the modes of interest and writes the result to gpurun_out/ and profiles/rNN_chain_compare.json:
    gpurun -- 'python profiles/chain_compare.py --round 02'"""
import argparse
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--round", default="02")
    ap.add_argument("--submaps", type=int, default=6)
    ap.add_argument("--scans", type=int, default=30)
    ap.add_argument("--racing-runs", type=int, default=4, help="repeat the default mode with the racing TSDF kernel")
    args = ap.parse_args()
    import torch
    from voxgraph_amd import capi
    from harness import chain_compare
    ctx = capi.Context(0)
    out = {}
    for name, kw in (("iso_mirrored_esdf", dict(use_esdf_distance=True, isosurface_points=True)),
                     ("iso_mirrored_tsdf", dict(use_esdf_distance=False, isosurface_points=True)),
                     ("voxels_esdf", dict(use_esdf_distance=True, isosurface_points=False))):
        out[name] = chain_compare.run(capi, ctx, torch, n_submaps=args.submaps, scans_per_submap=args.scans, **kw)
        r = out[name]
        print(name, {k: (round(v["xy_rmse_m"], 4) if isinstance(v, dict) and "xy_rmse_m" in v else None)
                     for k, v in r.items() if k.startswith("from_")})
    # run-to-run spread of the racing TSDF mode against the reproducible one
    spread = []
    for k in range(args.racing_runs):
        r = chain_compare.run(capi, ctx, torch, n_submaps=args.submaps, scans_per_submap=args.scans,
                              use_esdf_distance=True, isosurface_points=True, deterministic_tsdf=False)
        spread.append({k2: r[k2] for k2 in r if k2.startswith("from_")} | {"tsdf_gpu_vs_oracle": r["tsdf_gpu_vs_oracle"]})
        print("racing run", k, {k2: round(v["xy_rmse_m"], 4) for k2, v in r.items() if k2.startswith("from_") and "xy_rmse_m" in v})
    out["iso_mirrored_esdf_racing_runs"] = spread
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    for path in (os.path.join(ROOT, "gpurun_out", f"r{args.round}_chain_compare.json"),
                 os.path.join(ROOT, "profiles", f"r{args.round}_chain_compare.json")):
        json.dump(out, open(path, "w"), indent=1)
    ctx.close()


if __name__ == "__main__":
    main()
