"""ctypes binding of the C-ABI library (include/gsw_hip.h).

The product path has no CPU fallback: if `libgsw_hip.so` is missing or a symbol cannot be
resolved, importing this module's `lib()` raises.
"""
from __future__ import annotations

import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# GSW_HIP_LIBRARY points at an alternative build of the same library (A/B timing of kernel experiments)
LIB_PATH = os.environ.get("GSW_HIP_LIBRARY") or os.path.join(_HERE, "libgsw_hip.so")

RING_FP_M31 = 0
RING_ZPOW2 = 1
FLAG_NAND = 1
STATIC_SCRATCH_BYTES = 64 << 20  # GSW_STATIC_SCRATCH_BYTES

OK, E_INVALID, E_UNSUPPORTED, E_HIP, E_NOMEM = 0, 1, 2, 3, 4

_vp, _sz, _i, _u32, _u64, _dbl = C.c_void_p, C.c_size_t, C.c_int, C.c_uint32, C.c_uint64, C.c_double



class Gate(C.Structure):
    """gsw_gate of include/gsw_hip.h"""
    _fields_ = [("op", _u32), ("a", _u32), ("b", _u32), ("out", _u32)]


WIRE_CIPHERTEXT, WIRE_PUBLIC_KEY, WIRE_SECRET_KEY = 1, 2, 3
SAMPLER_NAIVE, SAMPLER_GAUSSIAN = 0, 1
DECOMP_U32, DECOMP_U8, DECOMP_BITS = 0, 1, 2
GATE_MULT, GATE_NAND, GATE_ADD, GATE_MULT_CONST = 0, 1, 2, 3

# every symbol include/gsw_hip.h declares: name -> (restype, argtypes)
SIGNATURES = {
    "gsw_ctx_create": (_i, [C.POINTER(_vp), _i, _i, _i, _i, _i]),
    "gsw_ctx_destroy": (None, [_vp]),
    "gsw_last_error": (C.c_char_p, []),
    "gsw_version": (C.c_char_p, []),
    "gsw_ctx_set_stream": (_i, [_vp, _vp]),
    "gsw_sync": (_i, [_vp]),
    "gsw_ctx_ell": (_i, [_vp]),
    "gsw_ctx_big_n": (_i, [_vp]),
    "gsw_ctx_ld": (_i, [_vp]),
    "gsw_ctx_gemm_variant": (_i, [_vp]),
    "gsw_ctx_compact_elems": (_sz, [_vp]),
    "gsw_ctx_prepared_bytes": (_sz, [_vp]),
    "gsw_ctx_rbits_elems": (_sz, [_vp]),
    "gsw_ctx_pk_prepared_bytes": (_sz, [_vp]),
    "gsw_ctx_set_workspace_limit": (_i, [_vp, _sz]),
    "gsw_ctx_enable_timing": (_i, [_vp, _i]),
    "gsw_ctx_kernel_time": (_i, [_vp, _i, C.POINTER(_dbl), C.POINTER(C.c_long)]),
    "gsw_ctx_reset_timing": (_i, [_vp]),
    "gsw_ctx_counters": (_i, [_vp, C.POINTER(_u64), C.POINTER(_u64)]),
    "gsw_malloc": (_i, [_vp, _sz, C.POINTER(_vp)]),
    "gsw_free": (_i, [_vp, _vp]),
    "gsw_memcpy_h2d": (_i, [_vp, _vp, _vp, _sz]),
    "gsw_memcpy_d2h": (_i, [_vp, _vp, _vp, _sz]),
    "gsw_memset_zero": (_i, [_vp, _vp, _sz]),
    "gsw_upload_compact": (_i, [_vp, _vp, _vp]),
    "gsw_download_compact": (_i, [_vp, _vp, _vp]),
    "gsw_bit_decomp": (_i, [_vp, _vp, _i, _i, _i, _vp]),
    "gsw_bit_decomp_row_elems": (_sz, [_vp, _i, _i]),
    "gsw_bit_decomp_as": (_i, [_vp, _vp, _i, _i, _i, _vp, _i]),
    "gsw_bit_decomp_inv": (_i, [_vp, _vp, _i, _i, _vp, _i, _i]),
    "gsw_flatten": (_i, [_vp, _vp, _i, _i]),
    "gsw_add": (_i, [_vp, _vp, _vp, _vp, _sz]),
    "gsw_mult_const": (_i, [_vp, _vp, _u32, _sz]),
    "gsw_mul_dense": (_i, [_vp, _vp, _vp, _vp, _i]),
    "gsw_nand_dense": (_i, [_vp, _vp, _vp, _vp, _i]),
    "gsw_ctx_set_dense_fast": (_i, [_vp, _i]),
    "gsw_ctx_last_dense_path": (_i, [_vp]),
    "gsw_ctx_last_combine": (_i, [_vp]),
    "gsw_validate_compact": (_i, [_vp, _vp, _sz, _i, C.POINTER(_u64)]),
    "gsw_prepare_rhs": (_i, [_vp, _vp, _sz, _vp, _i]),
    "gsw_mul_gadget_prepared": (_i, [_vp, _vp, _sz, _vp, _sz, _vp, _sz, _i, C.c_uint]),
    "gsw_mul_gadget_batch": (_i, [_vp, _vp, _sz, _vp, _sz, _vp, _sz, _i, C.c_uint]),
    "gsw_mul_gadget_batch_static": (_i, [_vp, _vp, _sz, _vp, _sz, _vp, _sz, _i, C.c_uint]),
    "gsw_mul_gadget_batch_digest": (_i, [_vp, _vp, _sz, _vp, _sz, _vp, _sz, _i, C.c_uint, _vp]),
    "gsw_eval_circuit": (_i, [_vp, _vp, _sz, _u32, C.POINTER(Gate), _sz]),
    "gsw_eval_circuit_cached": (_i, [_vp, _vp, _sz, _u32, _vp, _vp, C.POINTER(Gate), _sz]),
    "gsw_encrypt_batch_prepared": (_i, [_vp, _vp, _vp, _sz, _vp, _vp, _sz, _vp, _i]),
    "gsw_encrypt_batch_seeded": (_i, [_vp, _vp, _u64, _u64, _vp, _vp, _sz, _vp, _i]),
    "gsw_prepare_pk": (_i, [_vp, _vp, _vp]),
    "gsw_pack_rbits": (_i, [_vp, _vp, _vp]),
    "gsw_random_rbits": (_i, [_vp, _u64, _u64, _vp, _i]),
    "gsw_encrypt_batch": (_i, [_vp, _vp, _vp, _sz, _vp, _vp, _sz, _i]),
    "gsw_decrypt_batch": (_i, [_vp, _vp, _vp, _sz, _i, _vp, _vp]),
    "gsw_mp_decrypt_batch": (_i, [_vp, _vp, _vp, _sz, _i, _vp]),
    "gsw_partial_decrypt_batch": (_i, [_vp, _vp, _vp, _sz, _i, _vp]),
    "gsw_combine_partials": (_i, [_vp, _vp, _i, _i, _vp]),
    "gsw_matvec": (_i, [_vp, _vp, _i, _i, _i, _vp, _vp]),
    "gsw_decrypt_dense": (_i, [_vp, _vp, _vp, _vp]),
    "gsw_fill_uniform": (_i, [_vp, _u64, _u64, _u64, _u64, _i, _vp, _sz]),
    "gsw_sample_naive": (_i, [_vp, _u64, _u64, _vp, _sz]),
    "gsw_sample_gaussian": (_i, [_vp, _u64, _u64, _dbl, _vp, _sz]),
    "gsw_keygen_device": (_i, [_vp, _u64, _u64, _i, _dbl, _i, _vp, _vp, _vp]),
    "gsw_wire_bytes": (_sz, [_vp, _i, _i]),
    "gsw_wire_write_host": (_i, [_vp, _i, _vp, _vp, _sz]),
    "gsw_wire_read_host": (_i, [_vp, _i, _vp, _sz, _vp]),
    "gsw_wire_write_compact": (_i, [_vp, _vp, _sz, _i, _vp, _sz]),
    "gsw_wire_read_compact": (_i, [_vp, _vp, _sz, _vp, _sz, _i, C.POINTER(_i)]),
    "gsw_diag_stream_read": (_i, [_vp, _vp, _sz, _sz, _i, _sz]),
    "gsw_diag_limb_start_values": (_i, [_i, _i, _i, _i, _vp, C.POINTER(_i)]),
    "gsw_diag_bit_scale_mode": (_i, [_i, _i, _i]),
    "gsw_digest_batch": (_i, [_vp, _vp, _sz, _sz, _i, _vp]),
}

_lib = None


class GswError(RuntimeError):
    """A non-zero status from the engine (the Rust shim turns these into the reference's panics)."""

    def __init__(self, code: int, message: str):
        super().__init__(f"gsw error {code}: {message}")
        self.code = code
        self.message = message


def lib():
    """Load the HIP library; raises if it has not been built (there is no fallback path)."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise ImportError(
                f"{LIB_PATH} is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                "(the engine has no CPU fallback)")
        handle = C.CDLL(LIB_PATH)
        for name, (restype, argtypes) in SIGNATURES.items():
            fn = getattr(handle, name)  # AttributeError if the library lacks a declared symbol
            fn.restype = restype
            fn.argtypes = argtypes
        _lib = handle
    return _lib


def check(code: int) -> None:
    if code != OK:
        raise GswError(code, lib().gsw_last_error().decode())
