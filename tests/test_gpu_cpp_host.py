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


def test_plain_c_caller_agrees_with_the_python_mirror(oracle):
    """examples/c_caller.c drives the ABI from C99; its digests must be those of the oracle's products of the
    same seeded operands."""
    import re

    import numpy as np

    exe = os.path.join(ROOT, "examples", "c_caller")
    assert os.path.exists(exe), "build it with __graft_entry__.build()"
    n = 5
    res = subprocess.run([exe, str(n)], capture_output=True, text=True, timeout=120)
    assert res.returncode == 0, res.stdout + res.stderr
    m = re.search(r"N=(\d+) ld=(\d+) mult=([0-9a-f]{16}) nand=([0-9a-f]{16})", res.stdout)
    assert m, res.stdout
    big_n, ld = int(m.group(1)), int(m.group(2))
    ring = oracle.fp()
    raw = oracle.fill_uniform(ring, 7, 1, 0, 2147483646, 2 * big_n * ld).reshape(2, big_n, ld)
    c1, c2 = raw[0, :, :n + 1].copy(), raw[1, :, :n + 1].copy()
    og = oracle.GSW(ring, n, 31 * n)

    def padded_digest(x):  # the device digest runs over the padded N x ld layout
        full = np.zeros((big_n, ld), dtype=np.uint32)
        full[:, :n + 1] = x
        return oracle.digest(full.reshape(-1))

    assert int(m.group(3), 16) == padded_digest(og.mul_gadget(c1, c2))
    assert int(m.group(4), 16) == padded_digest(og.mul_gadget(c1, c2, flags=oracle.FLAG_NAND))
