"""Generates tests/golden/ref_reg_config1.npz from the REFERENCE's own cost function
(oracle/_ref/libref_reg.so = /root/reference's registration_cost_function.cpp compiled against
oracle/ref_shims).  Run here (needs /root/reference):

    python tests/golden/make_ref_golden.py

Inputs are the seeded config-1 pair (oracle/synth.py), rebuilt by the tests from the seed and
checked against the stored digests; outputs are stored for every `STRIDE`-th residual row plus a
SHA-256 of each complete output array (value-exactness check)."""
import hashlib
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import pyoracle as orc   # noqa: E402  (point extraction only)
from oracle import ref_reg, synth    # noqa: E402

STRIDE = 32
BASE = np.array([0.3, -0.2, 0.1, 0.05])
# (name, config kwargs, [(ref_pose, read_pose - ref_pose)], null blocks)
PERTURBATIONS = [np.array(p) for p in ([0, 0, 0, 0], [-0.3, 0.15, 0.0, 0.1], [0.15, -0.3, 0.15, -0.2],
                                       [0.3, 0.3, -0.3, 0.1], [0.05, -0.02, 0.01, 0.003])]


def digest(a):
    """SHA-256 of the values; `+ 0.0` maps -0.0 to +0.0 so that equal IEEE values hash equally
    (rows without correspondence are +0 in the reference, (-1)*(+0) in a kernel that forms
    J_read[:3] as -J_ref[:3])"""
    return hashlib.sha256(np.ascontiguousarray(np.asarray(a) + 0.0).tobytes()).hexdigest()


def inputs():
    ref, read = synth.config1_pair(seed=0, asymmetric=True)
    pts = {}
    for use_esdf in (True, False):
        pts[use_esdf] = orc.find_relevant_voxels(ref.voxel_size, ref.vps, ref.block_index,
                                                 ref.tsdf_distance, ref.tsdf_weight,
                                                 ref.esdf_distance if use_esdf else None, 1.0, 0.3)
    return ref, read, pts


CASES = [
    ("esdf_all", dict(use_esdf_distance=True), True),
    ("tsdf_nocorr", dict(use_esdf_distance=False, no_correspondence_cost=0.7), False),
    ("esdf_sampled", dict(use_esdf_distance=True, sampling_ratio=0.05), True),
]


def submap_scene():
    """tests/golden/ref_submap_scene.npz: the reference's finishSubmap() products, boxes and overlap
    list for the 7-submap scene of tests/test_ref_submap_pin.py (digests + block orders + boxes)."""
    from tests.test_ref_submap_pin import scene_submaps
    out, refs = {}, []
    for k, (i, sm, pose) in enumerate(scene_submaps()):
        R = ref_reg.Submap(i, pose, sm.voxel_size, sm.vps, sm.block_index, sm.tsdf_distance, sm.tsdf_weight,
                           sm.esdf_distance, sm.esdf_observed)
        refs.append(R)
        out[f"s{k}_id"] = np.int64(i)
        out[f"s{k}_pose"] = pose
        out[f"s{k}_block_order"] = R.block_order()
        for name, t in (("voxels", ref_reg.POINTS_VOXELS), ("iso", ref_reg.POINTS_ISOSURFACE)):
            xyz, d, w = R.points(t)
            out[f"s{k}_{name}_n"] = np.int64(len(w))
            out[f"s{k}_{name}_sha"] = np.array([digest(xyz), digest(d), digest(w)])
        out[f"s{k}_iso_blocks"] = np.unique(R.isosurface_blocks(), axis=0)
        out[f"s{k}_obb"] = np.concatenate(R.surface_obb())
        out[f"s{k}_aabb"] = np.concatenate(R.mission_surface_aabb())
    out["n_submaps"] = np.int64(len(refs))
    out["pairs"] = np.array([(a, b) for a in range(len(refs)) for b in range(a + 1, len(refs))
                             if refs[a].overlapsWith(refs[b])], np.int64)
    path = os.path.join(ROOT, "tests", "golden", "ref_submap_scene.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, os.path.getsize(path), "bytes;", len(out["pairs"]), "overlapping pairs")


def main():
    if not ref_reg.build():
        raise SystemExit("oracle/_ref/libref_reg.so could not be built (no /root/reference?)")
    submap_scene()
    ref, read, pts = inputs()
    out = {"stride": np.int64(STRIDE), "base_pose": BASE, "perturbations": np.array(PERTURBATIONS)}
    for name, arr in (("ref_tsdf", ref.tsdf_distance), ("ref_esdf", ref.esdf_distance),
                      ("read_tsdf", read.tsdf_distance), ("read_esdf", read.esdf_distance),
                      ("points_esdf_xyz", pts[True][0]), ("points_tsdf_d", pts[False][1])):
        out["input_sha_" + name] = np.array(digest(arr))
    for case, kw, use_esdf in CASES:
        xyz, d, w = pts[use_esdf]
        R = ref_reg.Submap(0, ref.pose, ref.voxel_size, ref.vps, ref.block_index, ref.tsdf_distance,
                           ref.tsdf_weight, ref.esdf_distance, ref.esdf_observed)
        R.set_points(ref_reg.POINTS_VOXELS, xyz, d, w)
        E = ref_reg.Submap(1, read.pose, read.voxel_size, read.vps, read.block_index,
                           read.tsdf_distance, read.tsdf_weight, read.esdf_distance, read.esdf_observed)
        cf = ref_reg.RegistrationCostFunction(R, E, ref_reg.POINTS_VOXELS, **kw)
        out[f"{case}_num_residuals"] = np.int64(cf.num_residuals())
        for k, pert in enumerate(PERTURBATIONS):     # successive calls: the sampler stream continues
            ok, r, j0, j1 = cf.Evaluate(BASE, BASE + pert)
            assert ok
            key = f"{case}_{k}"
            out[key + "_r"] = r[::STRIDE].copy()
            out[key + "_jref"] = j0[::STRIDE].copy()
            out[key + "_jread"] = j1[::STRIDE].copy()
            out[key + "_sha"] = np.array([digest(r), digest(j0), digest(j1)])
            out[key + "_corr"] = np.int64((np.abs(j0).sum(1) > 0).sum())
            out[key + "_cost"] = np.float64(0.5 * (r * r).sum())
    path = os.path.join(ROOT, "tests", "golden", "ref_reg_config1.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, os.path.getsize(path), "bytes")


if __name__ == "__main__":
    main()
