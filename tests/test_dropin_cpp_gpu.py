"""The drop-in, exercised the way a voxgraph maintainer would: ONE C++ process holds the reference's
own RegistrationCostFunction (compiled from /root/reference against oracle/ref_shims) and
voxgraph_amd::GpuRegistrationCostFunction, both constructed from the reference's own VoxgraphSubmap
objects (the GPU one through voxgraph_amd/cpp/voxgraph_submap_bridge.h), both called through
ceres::CostFunction::Evaluate, outputs compared entry by entry (oracle/ref_driver/dropin_check.cpp).

The binary is built by `make -C oracle ref` where /root/reference exists and travels to the GPU
box with the snapshot (oracle/_ref is git-ignored, not gpurun-ignored)."""
import os
import re
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BIN = os.path.join(ROOT, "oracle", "_ref", "dropin_check")

pytestmark = pytest.mark.gpu


@pytest.mark.skipif(not os.path.exists(BIN), reason="oracle/_ref/dropin_check not built (needs /root/reference)")
def test_gpu_cost_function_is_a_drop_in_for_the_reference_class():
    r = subprocess.run([BIN], capture_output=True, text=True, timeout=600)
    print(r.stdout[-2000:], r.stderr[-2000:])
    assert r.returncode == 0, (r.returncode, r.stdout[-2000:], r.stderr[-2000:])
    m = re.search(r"DROPIN cases=(\d+) failures=(\d+) values=(\d+) differing=(\d+) worst_abs=(\S+) worst_rel=(\S+)",
                  r.stdout)
    assert m, r.stdout
    cases, failures, values, differing = (int(m.group(i)) for i in range(1, 5))
    # 2 point types x 2 distance modes x {all points, 20 % sampled, 150 % sampled} x 3 successive evaluations
    assert cases == 36 and failures == 0 and values > 100000
    assert float(m.group(6)) <= 1e-4            # north_star tolerance
    assert differing == 0                        # and in fact every f64 is equal
