#!/usr/bin/env python3
"""Table of profiles/ab_headline.sh's outputs (gpurun_out/r05_ab) -> stdout; committed as profiles/r05_headline_ab.txt."""
import json
import os
import sys

d = sys.argv[1] if len(sys.argv) > 1 else "gpurun_out/r05_ab"


def load(name):
    p = os.path.join(d, name + ".json")
    if os.path.exists(p):
        return json.load(open(p))
    return json.loads(open(os.path.join(d, name + ".line")).read().strip().splitlines()[-1])


print("reg_eval_points_kernel<16,float,4>, config 3 (577.55 M evaluations per launch), 20 steps x 25 passes, HIP events")
print("same lease, same box, runs alternated: HEAD (bench.py) / round-2 tree (git archive 98ec687, its own bench.py + library)")
print(f"{'run':16s} {'kernel ms':>10s} {'frac':>6s} {'G eval/s':>9s}")
for n in ("head_1", "r02_1", "head_2", "r02_2", "head_3", "r02_3", "head_nocache", "head_noshipped", "default"):
    try:
        j = load(n)
    except (OSError, ValueError):
        continue
    rf = j["roofline"]
    print(f"{n:16s} {rf['kernel_ms']:10.4f} {rf['frac']:6.3f} {j['value'] / 1e3:9.1f}")
idle = json.load(open(os.path.join(d, "box_idle.json")))["snapshot"]
print("\nbox:", json.dumps({k: idle["sysfs"].get(k) for k in ("cards", "perf_level", "compute_partition", "memory_partition",
                                                              "power_cap_W", "mclk_levels", "sclk_levels", "vbios", "host_thp")}))
print("rocm-smi:", json.dumps(idle["rocm_smi"].get("card0")))
