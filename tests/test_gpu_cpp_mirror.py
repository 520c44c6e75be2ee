"""Builds (g++) and runs tests/cpp/test_host_mirror.cpp: the C++ host mirror (finmath-lib-cuda-extensions_amd/host/) against
the CPU twin through the same C++ interface — known answers of RandomVariableGPUTest.java:69-188 and its operator
differential test (:191-360) enforced bit-exactly, eager and fused."""
import os
import subprocess

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_cpp_host_mirror(fm, oracle, tmp_path):
    exe = str(tmp_path / "test_host_mirror")
    libdir = os.path.join(ROOT, "finmath-lib-cuda-extensions_amd", "lib")
    orcdir = os.path.join(ROOT, "oracle")
    subprocess.check_call(["g++", "-O1", "-std=c++17", "-ffp-contract=off", "-o", exe, os.path.join(ROOT, "tests", "cpp", "test_host_mirror.cpp"),
                           f"-L{libdir}", "-lfmhip", f"-L{orcdir}", "-lfm_oracle", f"-Wl,-rpath,{libdir}", f"-Wl,-rpath,{orcdir}", "-lm"])
    out = subprocess.run([exe], capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stdout + out.stderr
    assert out.stdout.strip().splitlines()[-1].startswith("OK")
