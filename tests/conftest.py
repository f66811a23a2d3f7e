import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu)")
    # A checkout without the built library (the .so is git-ignored): build it once, loudly, with
    # hipcc -- the same __graft_entry__.build() the driver runs.  Never a CPU substitute.
    lib = os.path.join(ROOT, "voxgraph_amd", "lib", "libvoxgraph_amd.so")
    if not os.path.exists(lib):
        import __graft_entry__
        __graft_entry__.build()


@pytest.fixture(scope="session")
def golden_dir():
    return os.path.join(ROOT, "tests", "golden")
