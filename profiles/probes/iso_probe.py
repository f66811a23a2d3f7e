#!/usr/bin/env python3
"""vgx_submap_extract_isosurface_points on a 256^3 city submap, five times (under rocprofv3 --kernel-trace --stats: the four
passes' kernel times; passes 1-3 run on the blocks pass 0 found candidates in only since round 6)"""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from voxgraph_amd import capi
capi.load()
ctx = capi.Context(0)
sm = capi.Submap.synth_city(ctx, 0, 0.2, 16, [-8, -8, -4], [16, 16, 16], 0.6, 2.0, 10.0, np.zeros(4), 2)
for rep in range(5):
    ctx.synchronize(); t0 = time.perf_counter(); n = sm.extract_isosurface_points(1.0); ctx.synchronize()
    print("isosurface 256^3: %.3f ms wall, %d points" % ((time.perf_counter() - t0) * 1e3, n))
