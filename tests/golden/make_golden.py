#!/usr/bin/env python3
"""Generates tests/golden/jacobians_xyz_yaw.json from the REFERENCE's own
symbolic derivation, /root/reference/voxgraph/scripts/jacobians_xyz_yaw.py.

The reference script is executed unmodified (runpy) in this container; its
symbolic T_eo (reference-submap frame -> reading-submap frame) is
differentiated with respect to the 8 pose parameters exactly as the script's
own print section does (jacobians_xyz_yaw.py:100-118) and evaluated at seeded
random poses/points.  /root/reference does not exist on the GPU box, so the
vectors are committed; re-run this script only to regenerate them.
"""
import contextlib
import io
import json
import os
import runpy

import numpy as np
import sympy as sp

REF = "/root/reference/voxgraph/scripts/jacobians_xyz_yaw.py"
OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "jacobians_xyz_yaw.json")


def main():
    with contextlib.redirect_stdout(io.StringIO()):
        ns = runpy.run_path(REF)
    T_eo, r = ns["T_eo"], ns["o_r_oi"]
    sym = {k: ns[k] for k in ("x_o", "y_o", "z_o", "theta_o", "x_e", "y_e", "z_e", "theta_e")}
    xi, yi, zi = ns["o_x_oi"], ns["o_y_oi"], ns["o_z_oi"]
    ref_params = [sym["x_o"], sym["y_o"], sym["z_o"], sym["theta_o"]]
    read_params = [sym["x_e"], sym["y_e"], sym["z_e"], sym["theta_e"]]
    M_ref = sp.Matrix.hstack(*[(sp.diff(T_eo, p) * r)[:3, :] for p in ref_params])
    M_read = sp.Matrix.hstack(*[(sp.diff(T_eo, p) * r)[:3, :] for p in read_params])
    p_read = (T_eo * r)[:3, :]
    args = ref_params + read_params + [xi, yi, zi]
    f_ref = sp.lambdify(args, M_ref, "numpy")
    f_read = sp.lambdify(args, M_read, "numpy")
    f_p = sp.lambdify(args, p_read, "numpy")
    rng = np.random.default_rng(20260925)
    cases = []
    for _ in range(32):
        ref_pose = np.concatenate([rng.uniform(-20, 20, 3), rng.uniform(-3.1, 3.1, 1)])
        read_pose = np.concatenate([rng.uniform(-20, 20, 3), rng.uniform(-3.1, 3.1, 1)])
        pt = rng.uniform(-25, 25, 3)
        vals = list(ref_pose) + list(read_pose) + list(pt)
        cases.append({
            "ref_pose": list(map(float, ref_pose)),
            "read_pose": list(map(float, read_pose)),
            "point": list(map(float, pt)),
            "M_ref": np.asarray(f_ref(*vals), float).tolist(),
            "M_read": np.asarray(f_read(*vals), float).tolist(),
            "p_read": np.asarray(f_p(*vals), float).ravel().tolist(),
        })
    with open(OUT, "w") as fh:
        json.dump({"source": "voxgraph/scripts/jacobians_xyz_yaw.py (run unmodified)",
                   "sympy": sp.__version__, "cases": cases}, fh, indent=1)
    print("wrote", OUT, len(cases), "cases")


if __name__ == "__main__":
    main()
