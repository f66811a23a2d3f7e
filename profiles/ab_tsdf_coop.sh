#!/bin/bash
# racing TSDF kernel, round 5: the cooperative kernel (vgx_tsdf_coop.hip) against the one-thread-per-point kernel
# (VGX_TSDF_KERNEL=v1) on one lease, and where its time goes: the same kernel with its last phases cut off
# (VGX_TSDF_ABLATE: 1 = no per-voxel folds, 2 = no walk either -- attribution runs only, the layer is wrong)
#   /usr/local/graft/bin/gpurun --timeout 900 -- bash profiles/ab_tsdf_coop.sh   -> profiles/r05_tsdf_racing.txt
OUT=gpurun_out/r05_tsdf
mkdir -p $OUT
P="timeout 100 python profiles/probes/tsdf_racing_probe.py"
VGX_TSDF_KERNEL=v1 $P > $OUT/v1.json 2> $OUT/v1.err
$P > $OUT/coop.json 2> $OUT/coop.err
VGX_PROBE_ORGANISED=1 $P > $OUT/coop_organised.json 2> $OUT/coop_organised.err
VGX_TSDF_ABLATE=1 $P > $OUT/coop_nofolds.json 2> $OUT/coop_nofolds.err
VGX_TSDF_ABLATE=2 $P > $OUT/coop_nowalk.json 2> $OUT/coop_nowalk.err
python - <<'PY'
import json
print("racing TSDF kernel by itself (HIP events around each scan's launch, stream drained before; 19 scans of the bench sessions)")
print(f"{'variant':26s} {'sensor':6s} {'one-point us':>12s} {'median us':>10s} {'min':>7s} {'max':>9s} {'back-to-back us':>16s}")
for n in ("v1", "coop", "coop_organised", "coop_nofolds", "coop_nowalk"):
    try:
        j = json.load(open(f"gpurun_out/r05_tsdf/{n}.json"))
    except Exception as e:
        print(n, "failed:", e)
        continue
    for k, v in j.items():
        if isinstance(v, dict):
            print(f"{n:26s} {k[:5]:6s} {v['one_point_scan_us']:12.1f} {v['kernel_us_median']:10.1f} {v['kernel_us_min']:7.1f} {v['kernel_us_max']:9.1f} {v['back_to_back_us']:16.1f}")
    if n in ("v1", "coop", "coop_organised"):
        for k, v in j.items():
            if isinstance(v, dict):
                print("    per scan:", k[:5], {a: round(b, 1) for a, b in v["per_scan"].items()})
                if v.get("trace"):
                    print("    workgroup stamps (counted scans, us):", {a: round(b, 2) for a, b in v["trace"].items()})
PY
