#!/usr/bin/env python3
"""bench.py's TSDF section by itself (both integrators, racing and reproducible modes):
    gpurun -- 'python profiles/tsdf_only.py'   or under rocprofv3 --kernel-trace --stats"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    import torch
    import bench
    from voxgraph_amd import capi
    capi.load()
    ctx = capi.Context(0)
    stream = torch.cuda.Stream()
    torch.cuda.set_stream(stream)
    ctx.set_stream(stream.cuda_stream)
    out = bench.tsdf_bench(capi, ctx, torch, cpu_scans=int(os.environ.get("CPU_SCANS", "8")))
    brief = {k: {"racing_ms": v["ms_per_scan"], "racing_kernel_ms": v["roofline"]["kernel_ms"],
                 "merged_ms": v["merged_integrator"]["ms_per_scan"], "merged_updates": v["merged_integrator"]["voxel_updates_per_scan"],
                 "merged_hbm_frac": v["merged_integrator"]["roofline"]["hbm_frac"],
                 "reproducible_ms": v["reproducible_mode"]["ms_per_scan"],
                 "reproducible_bit_identical": v["reproducible_mode"]["parity_vs_oracle"]["bit_identical"],
                 "updates": v["voxel_updates_per_scan"], "roofline": v["roofline"],
                 "cpu_1_core_Mpts": v["cpu_baseline"]["Mpoints_per_s"], "cpu_all_cores": v["cpu_baseline"]["all_cores"]}
             for k, v in out.items()}
    print(json.dumps(brief, indent=1))
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    json.dump(out, open(os.path.join(ROOT, "gpurun_out", os.environ.get("OUT_NAME", "tsdf_only.json")), "w"), indent=1)
    ctx.close()


if __name__ == "__main__":
    main()
