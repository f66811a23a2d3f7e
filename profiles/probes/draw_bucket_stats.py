#!/usr/bin/env python3
"""How long are the ranges the weighted draw's bucket table leaves to search?  (reg_draw_kernel: a range of <= 4
cumulative weights is four parallel loads, a longer one a dependent binary search the whole wavefront waits for.)
Isosurface points of a config-3 submap, the shipped sampling configuration's point sets.
    gpurun -- 'python profiles/probes/draw_bucket_stats.py'"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from voxgraph_amd import capi  # noqa: E402

capi.load()
ctx = capi.Context(0)
for k, pose in ((0, [0, 0, 0, 0.03]), (57, [250.0, 30.0, 0.0, -0.05])):
    sm = capi.Submap.synth_city(ctx, k, 0.2, 16, [-8, -8, -4], [16, 16, 16], 0.6, 2.0, 10.0, np.array(pose, float), 2)
    n = sm.extract_isosurface_points(1.0)
    xyz, dist, w = sm.download_points(capi.POINTS_ISOSURFACE)
    cum = np.cumsum(w.astype(np.float64))
    K = max(4096, 1 << int(np.ceil(np.log2(max(n, 1)))))
    edges = np.searchsorted(cum, np.arange(K + 1) / K * cum[-1], side="right")
    lens = np.diff(edges)
    print(f"submap {k}: n {n}, K {K}, weights min/median/max {w.min():.3g}/{np.median(w):.3g}/{w.max():.3g}, "
          f"distinct weights {len(np.unique(w))}")
    for thr in (1, 2, 4, 8, 16, 64):
        print(f"   buckets (= draws) with a range longer than {thr}: {(lens > thr).mean():.4f}")
    print("   longest range", lens.max(), " P(a wavefront of 64 lanes x 4 draws has one > 4):",
          1 - (1 - (lens > 4).mean()) ** 256)
    sm.destroy()
ctx.close()
