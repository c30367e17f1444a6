"""The C-ABI library loads and exports every symbol include/gsw_hip.h declares (no GPU needed)."""
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared_functions():
    text = open(os.path.join(ROOT, "include", "gsw_hip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(gsw_[a-z0-9_]+)\s*\(", text)))


def test_header_and_binding_agree():
    from tfhe_amd import _lib

    declared = _declared_functions()
    assert len(declared) >= 40
    assert sorted(_lib.SIGNATURES) == declared


def test_library_exports_every_declared_symbol():
    from tfhe_amd import _lib

    handle = _lib.lib()  # raises if the .so is missing or a symbol is absent
    for name in _declared_functions():
        assert hasattr(handle, name), name
    assert b"gfx950" in handle.gsw_version()


def test_no_gpu_means_a_loud_error_not_a_fallback():
    """Without a GPU the engine must refuse to work (there is no CPU path in the product)."""
    import torch

    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    import tfhe_amd

    with pytest.raises(tfhe_amd.GswError):
        tfhe_amd.Engine(tfhe_amd.Fp(), 10, 310)


def test_product_package_does_not_import_the_oracle():
    for dirpath, _, files in os.walk(os.path.join(ROOT, "tfhe_amd")):
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".cpp", ".hpp")):
                src = open(os.path.join(dirpath, f)).read()
                assert "oracle" not in src.replace("gsw_oracle", "oracle") or "import oracle" not in src, f
                assert "gsw_oracle" not in src and "libgsw_oracle" not in src, f


def test_header_is_plain_c(tmp_path):
    """include/gsw_hip.h must be consumable by a C compiler (cgo / bindgen / ctypes-style callers)."""
    import subprocess

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    src = tmp_path / "use_header.c"
    src.write_text('#include "gsw_hip.h"\n'
                   "int probe(gsw_ctx *ctx, gsw_gate *g) {\n"
                   "    g->op = GSW_GATE_NAND;\n"
                   "    return gsw_ctx_big_n(ctx) + (int)gsw_wire_bytes(ctx, GSW_WIRE_CIPHERTEXT, 1) + GSW_RING_FP_M31;\n"
                   "}\n")
    subprocess.run(["gcc", "-std=c99", "-pedantic", "-Wall", "-Werror", "-fsyntax-only", "-I",
                    os.path.join(root, "include"), str(src)], check=True)
