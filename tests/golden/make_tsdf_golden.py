#!/usr/bin/env python3
"""Golden digests of the TSDF oracle (oracle/tsdf_oracle.c, single thread, mixed order) on seeded scans:
    python tests/golden/make_tsdf_golden.py        -> tests/golden/tsdf_oracle_digests.json
The oracle restates voxblox from recall (parity unpinned: voxblox is not vendored in /root/reference), so
these digests pin the RESTATEMENT, not voxblox: tests/test_oracle_tsdf.py checks that the oracle still
produces them (a silent change of the restatement would move every GPU comparison with it), and
tests/test_tsdf_deterministic_gpu.py that the device's reproducible mode produces the same bytes."""
import hashlib
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
F = np.float32


def scan(n_az, n_el, seed, origin):
    rng = np.random.default_rng(seed)
    az = np.linspace(-np.pi, np.pi, n_az, endpoint=False) + rng.uniform(0, 1e-3)
    el = np.linspace(-0.35, 0.35, n_el)
    A, E = np.meshgrid(az, el)
    d = np.stack([np.cos(E) * np.cos(A), np.cos(E) * np.sin(A), np.sin(E)], -1).reshape(-1, 3)
    lo, hi = np.array([-5.0, -4.0, -1.0]) - origin, np.array([5.0, 4.0, 3.0]) - origin
    with np.errstate(divide="ignore", invalid="ignore"):
        t = np.where(d > 0, hi / d, np.where(d < 0, lo / d, np.inf)).min(1)
    pts = (d * t[:, None]).astype(F)
    pts[::53] *= F(3.0)                      # beyond the maximum range: clearing rays
    pts[7::61] *= F(0.005)                   # below the minimum range: dropped
    col = rng.integers(0, 256, (len(pts), 4)).astype(np.uint8)
    return pts, col


def sessions():
    """(name, config kwargs, merged?, [(T, points, colours)])"""
    out = []
    for name, kw, merged in (("fast_voxgraph_yaml", dict(default_truncation_distance=0.6, max_ray_length_m=8.0, use_const_weight=1,
                                                          use_weight_dropoff=1, use_sparsity_compensation_factor=1,
                                                          sparsity_compensation_factor=20.0), False),
                             ("fast_voxblox_defaults", dict(default_truncation_distance=0.4, max_ray_length_m=6.0), False),
                             ("merged_anti_grazing", dict(default_truncation_distance=0.6, max_ray_length_m=8.0,
                                                          enable_anti_grazing=1), True),
                             # integration_order_mode "sorted" (voxgraph_mapper.yaml:29): `integration_order` is the
                             # device's field (VGX_TSDF_ORDER_SORTED = 1); run() maps it to the oracle's (2)
                             ("fast_voxgraph_yaml_sorted_order", dict(default_truncation_distance=0.6, max_ray_length_m=8.0,
                                                                      use_const_weight=1, use_weight_dropoff=1,
                                                                      use_sparsity_compensation_factor=1,
                                                                      sparsity_compensation_factor=20.0, integration_order=1), False),
                             ("merged_sorted_order", dict(default_truncation_distance=0.6, max_ray_length_m=8.0,
                                                          integration_order=1), True)):
        scans = []
        for k in range(3):
            origin = np.array([0.3 * k - 0.2, -0.25 * k, 0.05 * k])
            pts, col = scan(512, 24, 100 + k, origin)
            yaw = 0.2 * k
            c, s_ = np.cos(-yaw), np.sin(-yaw)
            pts = np.stack([c * pts[:, 0] - s_ * pts[:, 1], s_ * pts[:, 0] + c * pts[:, 1], pts[:, 2]], 1).astype(F)
            T = np.r_[np.cos(yaw / 2), 0, 0, np.sin(yaw / 2), origin].astype(F)
            scans.append((T, pts, col))
        out.append((name, kw, merged, scans))
    return out


def oracle_kw(kw):
    """the sessions carry the DEVICE's integration_order (0 mixed, 1 sorted); the oracle's field counts
    0 = input order, 1 = mixed, 2 = sorted"""
    kw = dict(kw)
    kw["integration_order"] = {0: 1, 1: 2}[kw.get("integration_order", 0)]
    return kw


def digest(layer_download):
    h = hashlib.sha256()
    for a in layer_download:
        h.update(np.ascontiguousarray(a).tobytes())
    return h.hexdigest()


def run(make_layer, make_integrator, kw, merged, scans):
    layer = make_layer(0.2, 16)
    integ = make_integrator(kw, layer)
    updates = []
    for T, pts, col in scans:
        updates.append(int((integ.integratePointCloudMerged if merged else integ.integratePointCloud)(T, pts, col)))
    return {"updates": updates, "blocks": int(len(layer.download()[0])), "sha256": digest(layer.download())}


CFG_FIELDS = ("default_truncation_distance", "max_weight", "voxel_carving_enabled", "min_ray_length_m", "max_ray_length_m",
              "use_const_weight", "allow_clear", "use_weight_dropoff", "use_sparsity_compensation_factor",
              "sparsity_compensation_factor", "start_voxel_subsampling_factor", "max_consecutive_ray_collisions",
              "clear_checks_every_n_frames", "enable_anti_grazing", "integration_order")


def dump_sessions(path):
    """oracle/PIN.md: the sessions as one little-endian file for oracle/ref_driver/voxblox_tsdf_pin.cpp --
    int32 n_sessions; per session: char name[64], float cfg[15] (CFG_FIELDS, voxblox defaults where a session sets
    nothing; integration_order 0 mixed / 1 sorted), int32 merged, int32 n_scans; per scan: float T[7] (qw qx qy qz tx ty tz),
    int32 n, float pts[n][3], uint8 rgba[n][4].  Voxel size 0.2, 16 voxels per side."""
    import struct
    from oracle import pyoracle as orc
    with open(path, "wb") as f:
        ss = sessions()
        f.write(struct.pack("<i", len(ss)))
        for name, kw, merged, scans in ss:
            cfg = orc.tsdf_config(**oracle_kw(kw))
            vals = [float(getattr(cfg, k)) if k != "integration_order" else float(kw.get("integration_order", 0)) for k in CFG_FIELDS]
            f.write(name.encode().ljust(64, b"\0")[:64])
            f.write(struct.pack("<15f", *vals))
            f.write(struct.pack("<ii", int(merged), len(scans)))
            for T, pts, col in scans:
                f.write(np.asarray(T, F).tobytes())
                f.write(struct.pack("<i", len(pts)))
                f.write(np.ascontiguousarray(pts, F).tobytes())
                f.write(np.ascontiguousarray(col, np.uint8).tobytes())
    print("wrote", path)


def compare_layers(directory):
    """oracle/PIN.md: <directory>/<session>.layer.bin as voxblox_tsdf_pin wrote them (int32 n_blocks; per block int32
    index[3], float distance[4096], float weight[4096], uint8 rgba[4096][4]; any block order) against the oracle's
    layers, blocks matched by index.  Exit code 1 on the first session that differs."""
    from oracle import pyoracle as orc
    bad = 0
    for name, kw, merged, scans in sessions():
        layer = orc.TsdfLayer(0.2, 16)
        integ = orc.FastTsdfIntegrator(orc.tsdf_config(**oracle_kw(kw)), layer)
        for T, pts, col in scans:
            (integ.integratePointCloudMerged if merged else integ.integratePointCloud)(T, pts, col)
        bi, d, w, c = layer.download()
        raw = open(os.path.join(directory, name + ".layer.bin"), "rb").read()
        n = int(np.frombuffer(raw, np.int32, 1)[0])
        rec = np.dtype([("index", np.int32, 3), ("d", F, 4096), ("w", F, 4096), ("c", np.uint8, (4096, 4))])
        theirs = np.frombuffer(raw, rec, n, 4)
        ours = {tuple(b): k for k, b in enumerate(bi)}
        same_set = n == len(bi) and all(tuple(b) in ours for b in theirs["index"])
        nd = nw = nc = 0
        if same_set:
            for b in theirs:
                k = ours[tuple(b["index"])]
                nd += int((b["d"].view(np.uint32) != d[k].view(np.uint32)).sum())
                nw += int((b["w"].view(np.uint32) != w[k].view(np.uint32)).sum())
                nc += int((b["c"] != c[k].reshape(4096, 4)).any(1).sum())
        ok = same_set and nd == nw == nc == 0
        bad += 0 if ok else 1
        print(f"{name}: blocks voxblox {n} / oracle {len(bi)}, same block set {same_set}; voxels differing in distance {nd}, "
              f"weight {nw}, colour {nc} -> {'IDENTICAL' if ok else 'DIFFERENT'}")
    sys.exit(1 if bad else 0)


def main():
    if len(sys.argv) == 3 and sys.argv[1] == "--dump":
        return dump_sessions(sys.argv[2])
    if len(sys.argv) == 3 and sys.argv[1] == "--compare":
        return compare_layers(sys.argv[2])
    from oracle import pyoracle as orc
    out = {}
    for name, kw, merged, scans in sessions():
        out[name] = run(lambda vs, vps: orc.TsdfLayer(vs, vps),
                        lambda kw_, l: orc.FastTsdfIntegrator(orc.tsdf_config(**oracle_kw(kw_)), l), kw, merged, scans)
        print(name, out[name])
    json.dump(out, open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "tsdf_oracle_digests.json"), "w"), indent=1)


if __name__ == "__main__":
    main()
