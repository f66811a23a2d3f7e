#!/usr/bin/env python3
"""Per-call cost of the drop-in vgx_reg_evaluate on SMALL constraints (BASELINE config 1: a 64^3 pair, ~10^4 residuals; and
a few hundred residuals), where the call's fixed cost -- launch, D2H of 72 B per residual, completion wait -- is all there is;
the CPU port (oracle/reg_oracle.c, one core) beside it."""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import pyoracle as orc  # noqa: E402
from oracle import synth  # noqa: E402
from tests import helpers as H  # noqa: E402
from voxgraph_amd import capi  # noqa: E402

capi.load()
ctx = capi.Context(0)
sm, _ = synth.config1_pair()
g = H.gpu_submap(capi, ctx, sm)
xyz, dist, w = H.oracle_points(sm)
layer = H.oracle_layer(sm)
out = {}
for label, keep in (("config 1 (64^3 pair)", len(w)), ("2000 residuals", 2000), ("200 residuals", 200)):
    g.set_points(capi.POINTS_VOXELS, xyz[:keep], dist[:keep], w[:keep])
    cf = capi.RegistrationCostFunction(ctx, g, g, capi.default_config(registration_point_type=capi.POINTS_VOXELS))
    n = cf.num_residuals()
    r, jo, je = np.zeros(n), np.zeros((n, 4)), np.zeros((n, 4))
    a, b = np.array([0.02, -0.01, 0.03, 0.01]), np.array([0.07, 0.03, -0.02, 0.04])
    res = {}
    for what, jac in (("residual + 2 Jacobians", [jo, je]), ("residuals only", None)):
        for _ in range(20):
            cf.Evaluate([a, b], r, jac)
        t0, k = time.perf_counter(), 0
        while time.perf_counter() - t0 < 1.0:
            cf.Evaluate([a, b], r, jac)
            k += 1
        res[what] = {"us_per_evaluate_gpu": (time.perf_counter() - t0) / k * 1e6}
        t0, k = time.perf_counter(), 0
        while time.perf_counter() - t0 < 1.0:
            orc.reg_evaluate(layer, xyz[:keep], dist[:keep], w[:keep], a, b, want_jacobians=jac is not None) if "want_jacobians" in orc.reg_evaluate.__code__.co_varnames else orc.reg_evaluate(layer, xyz[:keep], dist[:keep], w[:keep], a, b)
            k += 1
        res[what]["us_per_evaluate_cpu_port_one_core"] = (time.perf_counter() - t0) / k * 1e6
    out[label] = {"residuals": int(n), **res}
    cf.destroy()
print(json.dumps(out, indent=1))
