#!/usr/bin/env python3
"""PCIe-inclusive rate of the drop-in entry point vgx_reg_evaluate (host f64 buffers in
Ceres layout) on one 256^3 pair of the bench scene.  Writes gpurun_out/dropin_pcie.json;
the committed copy is profiles/rNN_dropin_pcie.json.  Never used as bench `value`."""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from voxgraph_amd import capi  # noqa: E402

capi.load()
ctx = capi.Context(0)
poses = np.array([[0.0, 0, 0, 0.05], [25.6, 17.0, 0.0, -0.08]])
subs = [capi.Submap.synth_city(ctx, k, 0.2, 16, (-8, -8, -4), (16, 16, 16), 0.6, 2.0, 10.0,
                               poses[k], 2) for k in range(2)]
n = [s.extract_voxel_points(1.0, 0.3, True) for s in subs]
cf = capi.RegistrationCostFunction(ctx, subs[0], subs[1],
                                   capi.default_config(registration_point_type=capi.POINTS_VOXELS))
N = cf.num_residuals()
r, jo, je = np.zeros(N), np.zeros((N, 4)), np.zeros((N, 4))
est = poses + np.array([[0.2, 0.1, -0.1, 0.02], [0.0, -0.2, 0.1, -0.03]])
res = {}
for label, jac in (("residual+2jac", [jo, je]), ("residual+1jac(const block)", [None, je]),
                   ("residuals only", None)):
    cf.Evaluate([est[0], est[1]], r, jac)
    t0, k = time.perf_counter(), 0
    while time.perf_counter() - t0 < 2.0:
        cf.Evaluate([est[0], est[1]], r, jac)
        k += 1
    dt = (time.perf_counter() - t0) / k
    nbytes = 8 * N * (1 + (0 if jac is None else 4 * sum(j is not None for j in jac)))
    res[label] = {"ms_per_evaluate": dt * 1e3, "Mresiduals_per_s": N / dt / 1e6,
                  "D2H_GBs": nbytes / dt / 1e9}
out = {"what": "vgx_reg_evaluate (drop-in ceres::CostFunction::Evaluate), host pageable f64 buffers",
       "residuals": N, "results": res}
os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
json.dump(out, open(os.path.join(ROOT, "gpurun_out", "dropin_pcie.json"), "w"), indent=1)
print(json.dumps(out))
