"""Runs the C++ host-side mirror of `FheScheme` (include/gsw_hip.hpp) through the reference's own
unit tests on the GPU (tests/cpp/test_host_api.cpp, built by __graft_entry__.build())."""
import os
import subprocess

import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_cpp_host_api():
    exe = os.path.join(ROOT, "tests", "cpp", "test_host_api")
    assert os.path.exists(exe), "build it with __graft_entry__.build()"
    res = subprocess.run([exe], capture_output=True, text=True, timeout=600)
    assert res.returncode == 0, res.stdout + res.stderr
    assert "all host-API checks passed" in res.stdout
