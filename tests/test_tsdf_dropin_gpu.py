"""The TSDF drop-in the way a voxgraph maintainer would wire it: ONE C++ process integrates scans
with voxgraph_amd::GpuFastTsdfIntegrator into an UNBOUNDED GpuTsdfLayer, hands the layer to a
voxblox::Layer<TsdfVoxel> through voxgraph_amd/cpp/gpu_tsdf_layer_bridge.h (DownloadTsdfLayer /
UploadTsdfLayer; the stand-in voxblox headers of oracle/ref_shims) and compares it voxel for voxel
with the CPU restatement of voxblox's FastTsdfIntegrator (oracle/ref_driver/tsdf_dropin_check.cpp).
voxblox is not vendored in the reference: parity of the TSDF path stays UNPINNED; what this pins is
the boundary (layer hand-over, growth, no dropped updates) and GPU == oracle where the algorithm is
order-independent."""
import os
import re
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BIN = os.path.join(ROOT, "oracle", "_build", "tsdf_dropin_check")

pytestmark = pytest.mark.gpu


def test_gpu_integrator_and_layer_bridge_are_a_drop_in():
    if not os.path.exists(BIN):
        import __graft_entry__
        __graft_entry__.build()
    r = subprocess.run([BIN], capture_output=True, text=True, timeout=600)
    print(r.stdout[-2000:], r.stderr[-2000:])
    assert r.returncode == 0, (r.returncode, r.stdout[-2000:], r.stderr[-2000:])
    m = re.search(r"TSDF_DROPIN scans=(\d+) voxels=(\d+) differing=(\d+) blocks_cpu=(\d+) blocks_gpu=(\d+) growths=(\d+)",
                  r.stdout)
    assert m, r.stdout
    scans, voxels, differing, blocks_cpu, blocks_gpu, growths = (int(g) for g in m.groups())
    assert scans == 240 + 40 + 6 and voxels > 10 ** 6
    assert differing == 0 and blocks_cpu == blocks_gpu
    assert growths >= 3          # the sensor walked 60 m from a layer created without any box
