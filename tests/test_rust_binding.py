"""The `extern "C"` block of the Rust shim (integration/rust/hip.rs) against include/gsw_hip.h: every function the
shim binds must exist in the header with the same arity, the same C type for every parameter and the same return
type; the mirrored struct and constants must agree too.  No Rust toolchain exists here, so this textual check is
what stands between the shim and a silently mismatched ABI."""
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

TYPE_WORDS = {"int", "unsigned", "void", "char", "long", "double", "size_t", "uint32_t", "uint64_t", "uint8_t",
              "gsw_ctx", "gsw_gate", "const"}
RUST_SCALARS = {"c_int": "int", "c_uint": "unsigned", "c_long": "long", "c_char": "char", "c_void": "void",
                "usize": "size_t", "u32": "uint32_t", "u64": "uint64_t", "u8": "uint8_t", "f64": "double",
                "gsw_ctx": "gsw_ctx", "gsw_gate": "gsw_gate"}


def _c_type(decl: str) -> str:
    """Canonical spelling of a C parameter or return type: the parameter name dropped, one space between tokens."""
    tokens = re.findall(r"\w+|\*", decl)
    if len(tokens) > 1 and tokens[-1] != "*" and tokens[-1] not in TYPE_WORDS:
        tokens = tokens[:-1]
    return " ".join(tokens)


def _rust_type(t: str) -> str:
    t = t.strip()
    if t.startswith("*mut "):
        return _rust_type(t[5:]) + " *"
    if t.startswith("*const "):
        return "const " + _rust_type(t[7:]) + " *"
    return RUST_SCALARS[t]


def _header_prototypes():
    text = open(os.path.join(ROOT, "include", "gsw_hip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    with_defines = text
    text = re.sub(r"^\s*#.*$", "", text, flags=re.M)  # preprocessor lines are not part of any prototype
    protos = {}
    for ret, name, params in re.findall(r"([\w\s\*]+?)\b(gsw_[a-z0-9_]+)\s*\(([^)]*)\)\s*;", text):
        plist = [] if params.strip() in ("", "void") else [_c_type(p) for p in params.split(",")]
        protos[name] = (_c_type(ret + " x"), plist)
    return protos, with_defines


def _rust_externs():
    text = open(os.path.join(ROOT, "integration", "rust", "hip.rs")).read()
    block = re.search(r'extern "C" \{(.*?)\n\}', text, flags=re.S).group(1)
    fns = {}
    for name, params, ret in re.findall(r"fn (gsw_\w+)\((.*?)\)(?:\s*->\s*([^;]+))?;", block):
        plist = [_rust_type(p.split(":", 1)[1]) for p in params.split(",") if p.strip()]
        fns[name] = (_rust_type(ret) if ret else "void", plist)
    return fns, text


def test_every_bound_function_matches_the_header():
    protos, _ = _header_prototypes()
    fns, _ = _rust_externs()
    assert len(fns) >= 25
    for name, (ret, params) in fns.items():
        assert name in protos, f"{name} is bound by the shim but not declared in gsw_hip.h"
        c_ret, c_params = protos[name]
        assert ret == c_ret, f"{name}: returns {ret} in the shim, {c_ret} in the header"
        assert len(params) == len(c_params), f"{name}: {len(params)} parameters in the shim, {len(c_params)} in the header"
        for i, (r, c) in enumerate(zip(params, c_params)):
            assert r == c, f"{name}: parameter {i} is `{r}` in the shim, `{c}` in the header"


def test_the_parser_sees_a_mismatch():
    """The comparison is not vacuous: a doctored type is caught."""
    assert _rust_type("*mut *mut gsw_ctx") == "gsw_ctx * *" == _c_type("gsw_ctx **out")
    assert _rust_type("*const u32") == "const uint32_t *" == _c_type("const uint32_t *d_c1c")
    assert _rust_type("*const c_void") != _c_type("void *d_prepared")
    assert _c_type("unsigned flags") == "unsigned" and _c_type("size_t stride_out") == "size_t"


def test_struct_and_constants_agree():
    _, header = _header_prototypes()
    _, rust = _rust_externs()
    fields = re.search(r"typedef struct gsw_gate \{\s*uint32_t ([\w, ]+);\s*\} gsw_gate;", header).group(1)
    rust_fields = re.search(r"pub struct gsw_gate \{(.*?)\}", rust, flags=re.S).group(1)
    assert [f.strip() for f in fields.split(",")] == re.findall(r"pub (\w+): u32", rust_fields)
    assert "#[repr(C)]" in rust[:rust.index("pub struct gsw_gate")]
    for name, value in re.findall(r"pub const (GSW_\w+): \w+ = (\d+);", rust):
        m = re.search(rf"#define {name} (\d+)u?\b", header)
        assert m and m.group(1) == value, name
    # the ring kinds HipRing hands to gsw_ctx_create
    assert re.search(r"#define GSW_RING_FP_M31 0\b", header) and re.search(r"#define GSW_RING_ZPOW2 1\b", header)
    kinds = re.findall(r"const KIND: c_int = (\d);", rust)
    assert kinds == ["0", "1"]
