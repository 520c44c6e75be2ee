"""The engine's handle table (csrc/runtime.hpp: HandleTable) against a hash map, on the CPU: a C++ unit test compiled here."""
import os
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_handle_table_against_a_hash_map(tmp_path):
    gxx = shutil.which("g++")
    if not gxx or not os.path.isdir("/opt/rocm/include"):
        pytest.skip("no g++ / ROCm headers in this environment")
    exe = tmp_path / "test_handle_table"
    csrc = os.path.join(ROOT, "finmath-lib-cuda-extensions_amd", "csrc")
    subprocess.check_call([gxx, "-O1", "-std=c++17", "-Wall", "-D__HIP_PLATFORM_AMD__", "-I/opt/rocm/include", "-I", csrc, "-I", os.path.join(ROOT, "include"),
                           "-o", str(exe), os.path.join(ROOT, "tests", "cpp", "test_handle_table.cpp")])
    out = subprocess.run([str(exe)], capture_output=True, text=True, timeout=120)
    assert out.returncode == 0, out.stdout + out.stderr
    assert "handle table ok" in out.stdout
