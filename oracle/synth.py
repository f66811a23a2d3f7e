"""Synthetic submaps for tests and the bench's CPU leg (numpy, seeded).

TEST INFRASTRUCTURE.  The same analytic "city" scene is implemented on device
in voxgraph_amd/csrc/synth_scene.hip (bench tooling) and the two are compared
in tests/test_synth_scene.py, so the CPU oracle and the GPU path can be fed
identical grids at sizes where only one of them is practical to generate.

Value model: TSDF distance = clamp(d, +-trunc), weight 10 where
|d| <= 2*trunc else 0 (unobserved); ESDF = clamp(d, +-esdf_max), observed iff
|d| <= esdf_max.
"""
from dataclasses import dataclass

import numpy as np

F = np.float32


@dataclass
class SubmapData:
    voxel_size: float
    vps: int
    block_index: np.ndarray      # [n,3] int32
    tsdf_distance: np.ndarray    # [n,vps^3] f32
    tsdf_weight: np.ndarray      # [n,vps^3] f32
    esdf_distance: np.ndarray    # [n,vps^3] f32
    esdf_observed: np.ndarray    # [n,vps^3] u8
    pose: np.ndarray             # true (x,y,z,yaw) the grid was sampled at, f64

    @property
    def n_blocks(self):
        return self.block_index.shape[0]


# ----------------------------------------------------------------------------
# scenes: f(points[n,3] f32 world) -> signed distance f32
# ----------------------------------------------------------------------------
def plane_sdf(normal, offset):
    nrm = np.asarray(normal, F)

    def f(p):
        return (p @ nrm - F(offset)).astype(F)
    return f


def sphere_ground_sdf(centre, radius, ground_z):
    """BASELINE config 1: sphere + ground plane (union)."""
    c = np.asarray(centre, F)

    def f(p):
        ds = np.sqrt(((p - c) ** 2).sum(-1, dtype=F)).astype(F) - F(radius)
        dg = p[:, 2] - F(ground_z)
        return np.minimum(ds, dg).astype(F)
    return f


CITY_CELL = F(25.6)


def _fmix(h):
    h = h.astype(np.uint32)
    h ^= h >> np.uint32(16)
    h *= np.uint32(0x85EBCA6B)
    h ^= h >> np.uint32(13)
    h *= np.uint32(0xC2B2AE35)
    h ^= h >> np.uint32(16)
    return h


def _city_uniform(ci, cj, seed, k):
    """k-th uniform [0,1) f32 of cell (ci,cj); mirrored bit-for-bit on device."""
    with np.errstate(over="ignore"):
        h = (ci.astype(np.int32).view(np.uint32) * np.uint32(73856093)) ^ \
            (cj.astype(np.int32).view(np.uint32) * np.uint32(19349663)) ^ \
            (np.uint32(seed) * np.uint32(83492791))
        h = _fmix(h + np.uint32(k) * np.uint32(0x9E3779B9))
    return (h >> np.uint32(8)).astype(F) * F(1.0 / 16777216.0)


def city_sdf(seed):
    """Ground plane z=0 plus one box building per 25.6 m cell (config 3 scene)."""
    def f(p):
        p = p.astype(F)
        x, y, z = p[:, 0], p[:, 1], p[:, 2]
        ci0 = np.floor(x / CITY_CELL).astype(np.int32)
        cj0 = np.floor(y / CITY_CELL).astype(np.int32)
        d = z.copy()
        for di in (-1, 0, 1):
            for dj in (-1, 0, 1):
                ci = ci0 + np.int32(di)
                cj = cj0 + np.int32(dj)
                u0 = _city_uniform(ci, cj, seed, 0)
                u1 = _city_uniform(ci, cj, seed, 1)
                u2 = _city_uniform(ci, cj, seed, 2)
                u3 = _city_uniform(ci, cj, seed, 3)
                u4 = _city_uniform(ci, cj, seed, 4)
                cx = (ci.astype(F) + F(0.5)) * CITY_CELL + (u0 - F(0.5)) * F(6.0)
                cy = (cj.astype(F) + F(0.5)) * CITY_CELL + (u1 - F(0.5)) * F(6.0)
                hx = F(4.0) + F(5.0) * u2
                hy = F(4.0) + F(5.0) * u3
                top = F(6.0) + F(24.0) * u4
                cz = (top - F(10.0)) * F(0.5)
                hz = (top + F(10.0)) * F(0.5)
                qx = np.abs(x - cx) - hx
                qy = np.abs(y - cy) - hy
                qz = np.abs(z - cz) - hz
                ox = np.maximum(qx, F(0))
                oy = np.maximum(qy, F(0))
                oz = np.maximum(qz, F(0))
                outside = np.sqrt(ox * ox + oy * oy + oz * oz).astype(F)
                inside = np.minimum(np.maximum(qx, np.maximum(qy, qz)), F(0))
                d = np.minimum(d, outside + inside)
        return d.astype(F)
    return f


# ----------------------------------------------------------------------------
# grids
# ----------------------------------------------------------------------------
def voxel_centres(voxel_size, vps, block_index):
    """[n, vps^3, 3] f32 voxel centres in the submap frame, voxblox convention:
    origin = block * block_size, centre = origin + (i + 0.5) * voxel_size,
    linear index = x + vps * (y + vps * z)."""
    vs = F(voxel_size)
    bs = F(vps) * vs
    i = np.arange(vps, dtype=F)
    c = ((i + F(0.5)) * vs).astype(F)
    lz, ly, lx = np.meshgrid(c, c, c, indexing="ij")       # linear = x fastest
    local = np.stack([lx.ravel(), ly.ravel(), lz.ravel()], -1).astype(F)
    origin = (block_index.astype(F) * bs).astype(F)
    return (origin[:, None, :] + local[None, :, :]).astype(F)


def pose_apply(pose, p):
    """R(yaw) p + t in f32 (world = T_pose * submap point)."""
    c, s = F(np.cos(pose[3])), F(np.sin(pose[3]))
    x = c * p[..., 0] - s * p[..., 1] + F(pose[0])
    y = s * p[..., 0] + c * p[..., 1] + F(pose[1])
    z = p[..., 2] + F(pose[2])
    return np.stack([x, y, z], -1).astype(F)


def dense_block_index(block_min, block_dims):
    bx, by, bz = np.meshgrid(np.arange(block_dims[0]), np.arange(block_dims[1]),
                             np.arange(block_dims[2]), indexing="ij")
    bi = np.stack([bx.ravel(), by.ravel(), bz.ravel()], -1) + np.asarray(block_min)
    return bi.astype(np.int32)


def make_submap(sdf, voxel_size, vps, block_min, block_dims, trunc, pose=(0, 0, 0, 0),
                esdf_max=2.0, tsdf_weight=10.0, drop_empty_blocks=False, noise=0.0,
                seed=0):
    pose = np.asarray(pose, np.float64)
    bi = dense_block_index(block_min, block_dims)
    centres = voxel_centres(voxel_size, vps, bi)
    n = bi.shape[0]
    world = pose_apply(pose, centres.reshape(-1, 3))
    d = sdf(world).reshape(n, vps ** 3).astype(F)
    if noise > 0:
        rng = np.random.default_rng(seed)
        d = (d + rng.normal(0, noise * voxel_size, d.shape).astype(F)).astype(F)
    tsdf_d = np.clip(d, -F(trunc), F(trunc)).astype(F)
    tsdf_w = np.where(np.abs(d) <= F(2) * F(trunc), F(tsdf_weight), F(0)).astype(F)
    esdf_d = np.clip(d, -F(esdf_max), F(esdf_max)).astype(F)
    esdf_o = (np.abs(d) <= F(esdf_max)).astype(np.uint8)
    if drop_empty_blocks:
        keep = esdf_o.any(axis=1)
        bi, tsdf_d, tsdf_w, esdf_d, esdf_o = (a[keep] for a in
                                              (bi, tsdf_d, tsdf_w, esdf_d, esdf_o))
    return SubmapData(float(F(voxel_size)), vps, np.ascontiguousarray(bi),
                      np.ascontiguousarray(tsdf_d), np.ascontiguousarray(tsdf_w),
                      np.ascontiguousarray(esdf_d), np.ascontiguousarray(esdf_o), pose)


def union_sdf(*fs):
    def f(p):
        d = fs[0](p)
        for g in fs[1:]:
            d = np.minimum(d, g(p))
        return d.astype(F)
    return f


def sphere_sdf(centre, radius):
    c = np.asarray(centre, F)

    def f(p):
        return (np.sqrt(((p - c) ** 2).sum(-1, dtype=F)).astype(F) - F(radius)).astype(F)
    return f


def config1_pair(seed=0, asymmetric=False):
    """BASELINE config 1: two 64^3 submaps (4x4x4 blocks of 16^3, 0.10 m voxels),
    sphere r=2 m centred in the cube + ground plane, trunc 0.3 m, weight 10.
    The reading submap is a duplicate of the reference (test-bench design,
    registration_test_bench.cpp:178-185)."""
    vs, vps = 0.10, 16
    sdf = sphere_ground_sdf((3.2, 3.2, 3.2), 2.0, 0.45)
    if asymmetric:
        # sphere + plane is symmetric about the sphere's vertical axis (yaw about it is
        # unobservable); a second, smaller sphere makes the known-answer solve well posed
        sdf = union_sdf(sdf, sphere_sdf((1.3, 4.7, 1.2), 0.9))
    ref = make_submap(sdf, vs, vps, (0, 0, 0), (4, 4, 4), trunc=0.3, seed=seed)
    return ref, ref
