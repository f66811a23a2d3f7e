"""Saved map -> device submap (the device-side counterpart of VoxgraphSubmap::LoadFromStream,
voxgraph_submap.cpp:398-415, followed by finishSubmap as registration_test_bench.cpp:173-185 does)."""
import numpy as np
import pytest

from oracle import synth
from tests import helpers as H

pytestmark = pytest.mark.gpu


def test_collection_file_to_finished_device_submaps(tmp_path):
    from voxgraph_amd import capi
    ctx = capi.Context(0)
    ref, read = synth.config1_pair(seed=0, asymmetric=True)
    path = str(tmp_path / "pair.cblox")
    subs = [dict(id=3, T_M_S=[1, 0, 0, 0, 0, 0, 0], block_index=ref.block_index, tsdf_distance=ref.tsdf_distance,
                 tsdf_weight=ref.tsdf_weight, esdf_distance=ref.esdf_distance, esdf_observed=ref.esdf_observed),
            dict(id=4, T_M_S=[1, 0, 0, 0, 0, 0, 0], block_index=read.block_index, tsdf_distance=read.tsdf_distance,
                 tsdf_weight=read.tsdf_weight)]                      # TSDF only: ESDF regenerated on the device
    capi.write_map_file(path, capi.FILE_CBLOX_COLLECTION, ref.voxel_size, ref.vps, subs)
    f = capi.MapFile(path)
    a, b = f.load_submap(ctx, 0), f.load_submap(ctx, 1)
    b.generate_esdf()
    direct_a, direct_b = H.gpu_submap(capi, ctx, ref, 3), H.gpu_submap(capi, ctx, read, 4)
    direct_b.generate_esdf()
    for loaded, direct in ((a, direct_a), (b, direct_b)):
        assert loaded.extract_voxel_points(1.0, 0.3, True) == direct.extract_voxel_points(1.0, 0.3, True)
        assert loaded.extract_isosurface_points(1.0) == direct.extract_isosurface_points(1.0)
        for t in (capi.POINTS_VOXELS, capi.POINTS_ISOSURFACE):
            for x, y in zip(loaded.download_points(t), direct.download_points(t)):
                assert np.array_equal(x, y)
    cfg = capi.default_config(registration_point_type=capi.POINTS_VOXELS)
    pa, pb = np.array([0.1, 0.0, 0.05, 0.02]), np.array([0.0, 0.1, 0.0, -0.03])
    out = []
    for s0, s1 in ((a, b), (direct_a, direct_b)):
        cf = capi.RegistrationCostFunction(ctx, s0, s1, cfg)
        n = cf.num_residuals()
        r, j0, j1 = np.zeros(n), np.zeros((n, 4)), np.zeros((n, 4))
        assert cf.Evaluate([pa, pb], r, [j0, j1])
        out.append((r, j0, j1))
        cf.destroy()
    for x, y in zip(*out):
        assert np.array_equal(x, y)
    assert np.abs(out[0][0]).max() > 0
    for o in (a, b, direct_a, direct_b):
        o.destroy()
    f.close()
    ctx.close()
