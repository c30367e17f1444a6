"""Host-side checks of the contraction's number representation (no GPU).

The right operand is held as four signed base-256 digits per element and the accumulators of limb l start at
C_l instead of 0.  `gsw_diag_limb_start_values` returns the C_l the library uses; the digit split and the 32-bit
recombination of the kernels are restated here in numpy and checked against exact integer arithmetic, so that
the bounds the device code relies on are pinned by a test that runs everywhere.
"""
import ctypes as C

import numpy as np
import pytest

P = 2147483647
LEAN_LIMB_MAX = 8355000  # kLeanLimbMax of gsw_kernels.hip


def _start_values(ring_kind, bits, k_slots, wide):
    from tfhe_amd import _lib

    out = (C.c_uint32 * 4)()
    lean = C.c_int(-1)
    rc = _lib.lib().gsw_diag_limb_start_values(ring_kind, bits, k_slots, wide, out, C.byref(lean))
    return rc, [int(v) for v in out], lean.value


SCALE_WEIGHT = {0: 32, 1: 48, 2: 120}  # sum of the byte values 2^c over the 32 K slots of a word, per scaling mode


@pytest.mark.parametrize("k_slots", [128, 384, 8320, 16512, 16544, 32000, 32768, 65536, 1 << 19, 1 << 20])
@pytest.mark.parametrize("wide", [0, 1])
def test_fp_start_values_vanish_mod_p_and_bound_the_sums(k_slots, wide):
    from tfhe_amd import _lib

    rc, c, lean = _start_values(0, 31, k_slots, wide)
    assert rc == 0
    assert sum(v << (8 * l) for l, v in enumerate(c)) % P == 0
    mode = _lib.lib().gsw_diag_bit_scale_mode(0, 31, k_slots)
    assert mode in SCALE_WEIGHT
    units = k_slots // 32 * SCALE_WEIGHT[mode]  # a limb sum lies in [-128 units, 127 units]
    top = LEAN_LIMB_MAX if lean else 1 << 28
    for v in c:
        assert v - 128 * units >= 0 and v + 127 * units <= top
    if wide:
        assert lean == 0
    elif 255 * k_slots + 255 <= LEAN_LIMB_MAX:
        assert lean == 1  # every benchmark shape: n = 512 has 16 512 slots
    # the benchmark shapes: n = 256 scales by 2^(k & 3), n = 512 by 2^(k & 1), and both keep the 32-bit recombination
    assert _lib.lib().gsw_diag_bit_scale_mode(0, 31, 8320) == 2 and _lib.lib().gsw_diag_bit_scale_mode(0, 31, 16512) == 1
    assert _lib.lib().gsw_diag_bit_scale_mode(1, 32, 8320) == 0


def test_scaled_bits_identity():
    """(2^c bit) * (2^-c x) = bit * x mod p, and 2^-c x mod p is the 31-bit rotation the preparation uses."""
    rng = np.random.default_rng(11)
    x = np.concatenate([np.array(EDGES, np.uint64), rng.integers(0, P, 100000, dtype=np.uint64)])
    for c in range(4):
        rot = ((x >> np.uint64(c)) | (x << np.uint64(31 - c))) & np.uint64(P)
        assert np.array_equal((rot << np.uint64(c)) % np.uint64(P), x % np.uint64(P))
        assert rot.max() < P


def test_zpow2_start_values_vanish_mod_2_32():
    rc, c, lean = _start_values(1, 32, 16544, 0)
    assert rc == 0 and lean == 1
    assert sum(v << (8 * l) for l, v in enumerate(c)) % (1 << 32) == 0
    assert min(c) > 128 * 16544  # the sums of the n = 512 shapes stay positive


def test_bad_arguments_are_refused():
    from tfhe_amd import _lib

    lean = C.c_int()
    assert _lib.lib().gsw_diag_limb_start_values(0, 31, 8320, 0, None, C.byref(lean)) != 0
    assert _start_values(7, 31, 8320, 0)[0] != 0
    assert _start_values(0, 31, 0, 0)[0] != 0


def _signed_digits(x, fp):
    """signed_digits of gsw_kernels.hip: the int8 bytes of the prepared word, as integers in [-128, 127]."""
    x = x.astype(np.uint32).copy()
    if fp:
        x[x > 0x7F7F7F7F] -= np.uint32(P)
    y = (x + np.uint32(0x80808080)) ^ np.uint32(0x80808080)
    return np.stack([((y >> np.uint32(8 * l)) & np.uint32(0xFF)).astype(np.int64) for l in range(4)], axis=-1)


EDGES = [0, 1, 127, 128, 255, 256, 0x7F7F7F7F, 0x7F7F7F80, 0x7F7F7F81, 0x7F800000, 0x7FFFFF7F, 0x7FFFFF80, P - 2, P - 1]


def test_digits_represent_the_element():
    rng = np.random.default_rng(5)
    x = np.concatenate([np.array(EDGES, np.uint32), rng.integers(0, P, 200000, dtype=np.uint64).astype(np.uint32),
                        rng.integers(0x7F7F7F00, P, 20000, dtype=np.uint64).astype(np.uint32)])
    d = _signed_digits(x, fp=True)
    d = np.where(d >= 128, d - 256, d)
    assert d.min() >= -128 and d.max() <= 127
    val = sum(d[:, l] << (8 * l) for l in range(4))
    assert np.array_equal(val % P, x.astype(np.int64) % P)
    z = np.concatenate([np.array(EDGES + [0x80000000, 0x80808080, 0xFFFFFFFF, 0xFFFFFF80], np.uint32),
                        rng.integers(0, 1 << 32, 200000, dtype=np.uint64).astype(np.uint32)])
    d = _signed_digits(z, fp=False)
    d = np.where(d >= 128, d - 256, d)
    val = sum(d[:, l] << (8 * l) for l in range(4))
    assert np.array_equal(val % (1 << 32), z.astype(np.int64))


def _combine_lean_fp(a):
    """combine_limbs_lean<Fp> of gsw_kernels.hip in uint32 arithmetic."""
    a = a.astype(np.uint32)
    x = (a[:, 1] << np.uint32(8)) + a[:, 0]
    y = (a[:, 3] << np.uint32(8)) + a[:, 2]
    s = x + ((y & np.uint32(0x7FFF)) << np.uint32(16)) + (y >> np.uint32(15))
    f = (s & np.uint32(P)) + (s >> np.uint32(31))
    return np.minimum(f, f - np.uint32(P))


def test_lean_recombination_is_exact_up_to_its_bound():
    rng = np.random.default_rng(9)
    a = rng.integers(0, LEAN_LIMB_MAX + 1, size=(400000, 4), dtype=np.int64)
    corner = np.array([[LEAN_LIMB_MAX] * 4, [0] * 4, [LEAN_LIMB_MAX, 0, LEAN_LIMB_MAX, 0], [0, LEAN_LIMB_MAX, 0, LEAN_LIMB_MAX],
                       [P & 0xFF, (P >> 8) & 0xFF, (P >> 16) & 0xFF, P >> 24],  # the value p itself -> 0
                       [0, 0, 0x7FFF, 0], [LEAN_LIMB_MAX, LEAN_LIMB_MAX, 0x7FFF, LEAN_LIMB_MAX]], dtype=np.int64)
    a = np.concatenate([corner, a])
    with np.errstate(over="ignore"):
        got = _combine_lean_fp(a).astype(np.int64)
    want = (a[:, 0] + (a[:, 1] << 8) + (a[:, 2] << 16) + (a[:, 3] << 24)) % P
    assert np.array_equal(got, want)
    # the sum the fold sees never wraps 32 bits at the bound
    top = 257 * LEAN_LIMB_MAX
    assert top + (0x7FFF << 16) + (top >> 15) < 1 << 32
