"""`make -C oracle pin` (oracle/PIN.md: the one command for the day a voxblox / minkindr checkout is reachable) rehearsed on
a FAKE checkout laid out from the stand-in headers (oracle/pin_dryrun/README.md): the recipe's include order, its link
lines, ref_driver/voxblox_tsdf_pin.cpp, make_tsdf_golden.py --dump / --compare and the VGX_REF_DIR switch all execute.
Proves the recipe runs end to end -- nothing about voxblox (the fake voxblox wraps the oracle)."""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = "/root/reference/voxgraph/src/backend/constraint/cost_functions/registration_cost_function.cpp"


@pytest.mark.skipif(not os.path.exists(REF), reason="the reference sources are not on this host (the recipe compiles them)")
def test_the_pin_recipe_executes_end_to_end_on_a_fake_checkout():
    out = subprocess.run(["make", "-C", os.path.join(ROOT, "oracle"), "pin-dryrun"], capture_output=True, text=True, timeout=900)
    log = out.stdout + out.stderr
    assert out.returncode == 0, log[-3000:]
    assert os.path.exists(os.path.join(ROOT, "oracle", "_ref_pinned_dryrun", "libref_reg.so"))
    assert os.path.exists(os.path.join(ROOT, "oracle", "_ref_pinned_dryrun", "voxblox_tsdf_pin"))
    # the pin tests ran against the re-pinned library, and every session's layer was compared voxel for voxel
    assert " passed" in log and "failed" not in log
    assert log.count("-> IDENTICAL") == 5 and "DIFFERENT" not in log, log[-2000:]
