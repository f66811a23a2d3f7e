"""The C++ host-side mirrors (voxgraph_amd/cpp/*.h) compile against a Ceres stub on CPU
and, on a GPU box, reproduce closed forms through the C ABI from plain C++."""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
INC = ["-I", os.path.join(ROOT, "include"), "-I", os.path.join(ROOT, "voxgraph_amd", "cpp"),
       "-I", os.path.join(ROOT, "tests", "stubs")]
SRC = os.path.join(ROOT, "tests", "cpp", "adapter_smoke.cpp")
SOLVE_SRC = os.path.join(ROOT, "tests", "cpp", "solve_smoke.cpp")


def _build(tmp_path, src=SRC, name="adapter_smoke"):
    import __graft_entry__ as g
    g.build()
    exe = str(tmp_path / name)
    lib = os.path.join(ROOT, "voxgraph_amd", "lib")
    subprocess.check_call(["g++", "-std=c++14", "-O1", "-Wall", *INC, src, "-o", exe, "-L", lib,
                           "-lvoxgraph_amd", "-lpthread", f"-Wl,-rpath,{lib}", "-Wl,-rpath,/opt/rocm/lib"])
    return exe


def test_cpp_mirrors_compile_and_link_against_the_c_abi(tmp_path):
    """C++14 like the reference (voxgraph/CMakeLists.txt:4); no GPU needed to build."""
    assert os.path.exists(_build(tmp_path))


@pytest.mark.gpu
def test_cpp_mirrors_run_on_gpu(tmp_path):
    out = subprocess.run([_build(tmp_path)], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0 and "ADAPTER_SMOKE_OK" in out.stdout, out.stdout + out.stderr


def test_cpp_solve_compiles(tmp_path):
    assert os.path.exists(_build(tmp_path, SOLVE_SRC, "solve_smoke"))


@pytest.mark.gpu
def test_cpp_solve_ends_in_the_same_pose_on_every_integration_route(tmp_path):
    """ceres::Problem + ceres::Solve (Ceres-shaped stub) over the drop-in cost functions, the batched
    evaluation callback and the two-context multi-GPU callback: same end pose within 1 mm / 0.01 deg."""
    out = subprocess.run([_build(tmp_path, SOLVE_SRC, "solve_smoke")], capture_output=True, text=True, timeout=300)
    print(out.stdout)
    assert out.returncode == 0 and "SOLVE_SMOKE_OK" in out.stdout, out.stdout + out.stderr
