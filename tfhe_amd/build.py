"""Builds libgsw_hip.so in-tree with hipcc for gfx950 (cross-compiles without a GPU)."""
from __future__ import annotations

import os
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
SOURCES = [os.path.join(HERE, "csrc", "gsw_kernels.hip"), os.path.join(HERE, "csrc", "gsw_capi.hip"),
           os.path.join(HERE, "csrc", "gsw_wire.cpp")]  # gsw_wire.cpp: host-only (wire format parser, sampler table)
HEADERS = [os.path.join(HERE, "csrc", "gsw_kernels.h"), os.path.join(HERE, "csrc", "gsw_wire.h"),
           os.path.join(HERE, "..", "include", "gsw_hip.h")]
# -DGSW_DIAGNOSTIC_BUILDS additionally compiles csrc/gsw_kernels_diag.inc (superseded kernels, for A/B timing),
# -DGSW_EXP_TALL csrc/gsw_kernels_tall.inc (the tall-tile experiment), -DGSW_EXP_WIDE8=<depth> / -DGSW_EXP_PERSIST
# csrc/gsw_kernels_exp.inc (eight-wave tiles, persistent grids); the shipped library holds none of them
OUT = os.path.join(HERE, "libgsw_hip.so")


def needs_build() -> bool:
    if not os.path.exists(OUT):
        return True
    t = os.path.getmtime(OUT)
    return any(os.path.getmtime(p) > t for p in SOURCES + HEADERS)


def build(force: bool = False) -> str:
    if force or needs_build():
        hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
        cmd = [hipcc, "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-Wno-unused-value",
               "-o", OUT] + SOURCES + ["-ldl"]
        subprocess.check_call(cmd)
    return OUT


if __name__ == "__main__":
    print(build(force=True))
