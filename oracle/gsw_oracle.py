"""ctypes front end of the CPU oracle (oracle/gsw_oracle.c).

TEST INFRASTRUCTURE ONLY: importable from tests/, __graft_entry__.smoke() and the
cpu_baseline leg of bench.py, never from the product package.  All matrices are
row-major numpy uint32 arrays of canonical values in [0, q).

Parity status: the Rust reference cannot be built here; the oracle is pinned by the
reference's own known-answer tests only (see gsw_oracle.h).  mult/add/mult_const/nand
are "parity unpinned": the reference holds no vectors for them.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess
from dataclasses import dataclass

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "libgsw_oracle.so")

RING_FP = 0
RING_ZPOW2 = 1
FP_P = 2147483647
FLAG_NAND = 1


def build(force: bool = False) -> str:
    """Compile the oracle with gcc (oracle/Makefile)."""
    src = os.path.join(_HERE, "gsw_oracle.c")
    if force or not os.path.exists(_LIB_PATH) or os.path.getmtime(_LIB_PATH) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _HERE, "-s", "libgsw_oracle.so"])
    return _LIB_PATH


class _Ring(C.Structure):
    _fields_ = [("kind", C.c_int), ("ell", C.c_int), ("mask", C.c_uint32)]


_lib = None


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(_LIB_PATH):
            build()
        _lib = C.CDLL(_LIB_PATH)
        _lib.gswo_from_u64.restype = C.c_uint32
        _lib.gswo_from_u64.argtypes = [C.c_void_p, C.c_uint64]
        for name in ("gswo_add", "gswo_sub", "gswo_mul"):
            getattr(_lib, name).restype = C.c_uint32
            getattr(_lib, name).argtypes = [C.c_void_p, C.c_uint32, C.c_uint32]
        _lib.gswo_neg.restype = C.c_uint32
        _lib.gswo_neg.argtypes = [C.c_void_p, C.c_uint32]
        _lib.gswo_max_u64.restype = C.c_uint64
        _lib.gswo_max_u64.argtypes = [C.c_void_p]
        _lib.gswo_fp_inv.restype = C.c_uint32
        _lib.gswo_fp_inv.argtypes = [C.c_uint32]
        _lib.gswo_is_zero_one.restype = C.c_uint32
        _lib.gswo_is_zero_one.argtypes = [C.c_void_p, C.c_uint32]
        _lib.gswo_dot.restype = C.c_uint32
        _lib.gswo_decrypt.restype = C.c_uint32
        _lib.gswo_decrypt_compact.restype = C.c_uint32
        _lib.gswo_compact_row_dot_s.restype = C.c_uint32
        _lib.gswo_rng_u64.restype = C.c_uint64
        _lib.gswo_rng_u64.argtypes = [C.c_uint64, C.c_uint64, C.c_uint64]
        _lib.gswo_digest.restype = C.c_uint64
    return _lib


def _p(a: np.ndarray):
    return a.ctypes.data_as(C.c_void_p)


def _u32(a) -> np.ndarray:
    return np.ascontiguousarray(a, dtype=np.uint32)


@dataclass
class Ring:
    """R: RingElement (lib.rs:15-29) as (kind, Num_Bits)."""

    kind: int
    log2q: int = 31

    def __post_init__(self):
        self._c = _Ring()
        if lib().gswo_ring_init(C.byref(self._c), self.kind, self.log2q) != 0:
            raise ValueError(f"unsupported ring kind={self.kind} log2q={self.log2q}")
        self.ell = int(self._c.ell)
        self.mask = int(self._c.mask)
        self.q = FP_P if self.kind == RING_FP else (1 << self.log2q)

    @property
    def ref(self):
        return C.byref(self._c)

    # element ops
    def from_u64(self, x: int) -> int:
        return int(lib().gswo_from_u64(self.ref, C.c_uint64(x & 0xFFFFFFFFFFFFFFFF)))

    def add(self, a, b):
        return int(lib().gswo_add(self.ref, a, b))

    def sub(self, a, b):
        return int(lib().gswo_sub(self.ref, a, b))

    def mul(self, a, b):
        return int(lib().gswo_mul(self.ref, a, b))

    def neg(self, a):
        return int(lib().gswo_neg(self.ref, a))

    def max_u64(self) -> int:
        return int(lib().gswo_max_u64(self.ref))

    def is_zero_one(self, x: int) -> int:
        return int(lib().gswo_is_zero_one(self.ref, x))

    def big_n(self, n: int) -> int:
        return self.ell * (n + 1)


def fp() -> Ring:
    return Ring(RING_FP, 31)


def zpow2(m: int) -> Ring:
    return Ring(RING_ZPOW2, m)


def fp_inv(a: int) -> int:
    return int(lib().gswo_fp_inv(a))


# ---- linear algebra -------------------------------------------------------
def matmul(r: Ring, a, b) -> np.ndarray:
    a, b = _u32(a), _u32(b)
    rows, inner = a.shape
    inner2, cols = b.shape
    assert inner == inner2
    out = np.empty((rows, cols), dtype=np.uint32)
    lib().gswo_matmul(r.ref, _p(a), _p(b), _p(out), C.c_size_t(rows), C.c_size_t(inner), C.c_size_t(cols))
    return out


def dot(r: Ring, a, b) -> int:
    a, b = _u32(a), _u32(b)
    assert a.shape == b.shape
    return int(lib().gswo_dot(r.ref, _p(a), _p(b), C.c_size_t(a.size)))


# ---- gadget helpers ---------------------------------------------------------
def gadget_vector(r: Ring) -> np.ndarray:
    g = np.empty(r.ell, dtype=np.uint32)
    lib().gswo_gadget_vector(r.ref, _p(g))
    return g


def bit_decomp_matrix(r: Ring, a) -> np.ndarray:
    a = _u32(a)
    rows, cols = a.shape
    out = np.empty((rows, cols * r.ell), dtype=np.uint32)
    lib().gswo_bit_decomp_matrix(r.ref, _p(a), _p(out), C.c_size_t(rows), C.c_size_t(cols))
    return out


def bit_decomp_inv_matrix(r: Ring, a, quirk: bool = True) -> np.ndarray:
    a = _u32(a)
    rows, wide = a.shape
    assert wide % r.ell == 0
    cols = wide // r.ell
    out = np.empty((rows, cols), dtype=np.uint32)
    lib().gswo_bit_decomp_inv_matrix(r.ref, _p(a), _p(out), C.c_size_t(rows), C.c_size_t(cols), int(bool(quirk)))
    return out


def flatten_matrix(r: Ring, a) -> np.ndarray:
    a = _u32(a).copy()
    rows, wide = a.shape
    lib().gswo_flatten_matrix(r.ref, _p(a), C.c_size_t(rows), C.c_size_t(wide))
    return a


def powers_of_2(r: Ring, b) -> np.ndarray:
    b = _u32(b)
    out = np.empty(b.size * r.ell, dtype=np.uint32)
    lib().gswo_powers_of_2(r.ref, _p(b), C.c_size_t(b.size), _p(out))
    return out


# ---- keys ---------------------------------------------------------------------
@dataclass
class SecretKey:
    """GswSk (gsw/sk.rs:13-17)."""

    t: np.ndarray
    s: np.ndarray
    v: np.ndarray


@dataclass
class PublicKey:
    """GswPk (gsw/pk.rs:13-16)."""

    b: np.ndarray
    pk_matrix: np.ndarray


def sk_new(r: Ring, t) -> SecretKey:
    t = _u32(t)
    n = t.size
    s = np.empty(n + 1, dtype=np.uint32)
    v = np.empty((n + 1) * r.ell, dtype=np.uint32)
    lib().gswo_sk_new(r.ref, _p(t), C.c_size_t(n), _p(s), _p(v))
    return SecretKey(t, s, v)


def pk_new(r: Ring, rand_b, e, t) -> PublicKey:
    rand_b, e, t = _u32(rand_b), _u32(e), _u32(t)
    m, n = rand_b.shape
    if n != t.size:
        raise ValueError(f"Dimension Mismatch ncols {n}, {t.size}")  # pk.rs:24
    if m != e.size:
        raise ValueError(f"Dimension Mismatch nrows {m}, {e.size}")  # pk.rs:25
    b = np.empty(m, dtype=np.uint32)
    pk = np.empty((m, n + 1), dtype=np.uint32)
    lib().gswo_pk_new(r.ref, _p(rand_b), _p(e), _p(t), C.c_size_t(m), C.c_size_t(n), _p(b), _p(pk))
    return PublicKey(b, pk)


# ---- sampling -------------------------------------------------------------------
def rng_u64(seed: int, stream: int, counter: int) -> int:
    return int(lib().gswo_rng_u64(seed, stream, counter))


def fill_uniform(r: Ring, seed: int, stream: int, lo: int, hi: int, count: int, checked: bool = True) -> np.ndarray:
    out = np.empty(count, dtype=np.uint32)
    if checked:
        rc = lib().gswo_fill_uniform(r.ref, C.c_uint64(seed), C.c_uint64(stream), C.c_uint64(lo), C.c_uint64(hi),
                                     _p(out), C.c_size_t(count))
        if rc != 0:
            raise AssertionError("assertion failed: max <= P")  # error_sampling.rs:21
    else:
        lib().gswo_fill_uniform_unchecked(r.ref, C.c_uint64(seed), C.c_uint64(stream), C.c_uint64(lo),
                                          C.c_uint64(hi), _p(out), C.c_size_t(count))
    return out


def fill_bits(seed: int, stream: int, count: int) -> np.ndarray:
    out = np.empty(count, dtype=np.uint8)
    lib().gswo_fill_bits(C.c_uint64(seed), C.c_uint64(stream), _p(out), C.c_size_t(count))
    return out


def sample_naive(r: Ring, seed: int, stream: int, count: int) -> np.ndarray:
    out = np.empty(count, dtype=np.uint32)
    lib().gswo_sample_naive(r.ref, C.c_uint64(seed), C.c_uint64(stream), _p(out), C.c_size_t(count))
    return out


def sample_gaussian(r: Ring, seed: int, stream: int, sigma: float, count: int) -> np.ndarray:
    out = np.empty(count, dtype=np.uint32)
    lib().gswo_sample_gaussian(r.ref, C.c_uint64(seed), C.c_uint64(stream), C.c_double(sigma), _p(out),
                               C.c_size_t(count))
    return out


def sample_gaussian_i32(seed: int, stream: int, sigma: float, count: int) -> np.ndarray:
    out = np.empty(count, dtype=np.int32)
    lib().gswo_sample_gaussian_i32(C.c_uint64(seed), C.c_uint64(stream), C.c_double(sigma), _p(out),
                                   C.c_size_t(count))
    return out


def digest(x) -> int:
    x = _u32(x).reshape(-1)
    return int(lib().gswo_digest(_p(x), C.c_size_t(x.size)))


# ---- the scheme (gsw/gsw_impl.rs) ------------------------------------------------
class GSW:
    """CPU restatement of `impl FheScheme<R> for GSW<R, T>` (gsw/gsw_impl.rs:12-125).

    `sampler` is "naive" (NaiveSampler), "gaussian" (DiscrGaussianSampler::default, sigma = 1)
    or a float sigma.  Randomness comes from the oracle's counter-based generator keyed by
    `seed`; the reference uses the OS-seeded thread rng, so streams cannot agree.
    """

    def __init__(self, ring: Ring, n: int, m: int, sampler="gaussian", seed: int = 1, check_max: bool = True):
        self.ring, self.n, self.m, self.sampler, self.seed = ring, int(n), int(m), sampler, int(seed)
        self.check_max = check_max
        self._draw = 0

    @property
    def big_n(self) -> int:
        return self.ring.big_n(self.n)

    def _stream(self) -> int:
        self._draw += 1
        return self._draw

    def _errors(self, count: int) -> np.ndarray:
        if self.sampler == "naive":
            return sample_naive(self.ring, self.seed, self._stream(), count)
        sigma = 1.0 if self.sampler == "gaussian" else float(self.sampler)
        return sample_gaussian(self.ring, self.seed, self._stream(), sigma, count)

    def keygen(self):
        """gsw_impl.rs:17-25."""
        r = self.ring
        mx = r.max_u64()
        t = fill_uniform(r, self.seed, self._stream(), 0, mx, self.n, checked=self.check_max)
        sk = sk_new(r, t)
        err = self._errors(self.m)
        rand_b = fill_uniform(r, self.seed, self._stream(), 0, mx, self.m * self.n,
                              checked=self.check_max).reshape(self.m, self.n)
        return sk, pk_new(r, rand_b, err, t)

    def random_bits(self) -> np.ndarray:
        """rnd_dmatrix(big_n, m, 0, 1) of gsw_impl.rs:30, as a 0/1 byte matrix."""
        return fill_bits(self.seed, self._stream(), self.big_n * self.m).reshape(self.big_n, self.m)

    def encrypt(self, pk: PublicKey, message: int, rbits=None) -> np.ndarray:
        """gsw_impl.rs:27-42 (default features): dense N x N."""
        r = self.ring
        rbits = self.random_bits() if rbits is None else np.ascontiguousarray(rbits, dtype=np.uint8)
        pkm = _u32(pk.pk_matrix)
        out = np.empty((self.big_n, self.big_n), dtype=np.uint32)
        lib().gswo_encrypt(r.ref, C.c_size_t(self.n), C.c_size_t(self.m), _p(pkm), _p(rbits), C.c_uint32(message),
                           _p(out))
        return out

    def encrypt_compact(self, pk: PublicKey, message: int, rbits=None) -> np.ndarray:
        r = self.ring
        rbits = self.random_bits() if rbits is None else np.ascontiguousarray(rbits, dtype=np.uint8)
        pkm = _u32(pk.pk_matrix)
        out = np.empty((self.big_n, self.n + 1), dtype=np.uint32)
        lib().gswo_encrypt_compact(r.ref, C.c_size_t(self.n), C.c_size_t(self.m), _p(pkm), _p(rbits),
                                   C.c_uint32(message), _p(out))
        return out

    def decrypt(self, sk: SecretKey, c) -> int:
        """gsw_impl.rs:48-52."""
        c = _u32(c)
        assert c.shape == (self.big_n, self.big_n)
        return int(lib().gswo_decrypt(self.ring.ref, C.c_size_t(self.n), _p(_u32(sk.v)), _p(c)))

    def mp_decrypt(self, sk: SecretKey, c) -> int:
        """gsw_impl.rs:67-84."""
        c = _u32(c)
        out = C.c_uint32(0)
        rc = lib().gswo_mp_decrypt(self.ring.ref, C.c_size_t(self.n), _p(_u32(sk.v)), _p(c), C.byref(out))
        if rc != 0:
            raise RuntimeError("Only supported for pow2 rings!")  # gsw_impl.rs:69
        return int(out.value)

    def add(self, c1, c2) -> np.ndarray:
        """gsw_impl.rs:87-94."""
        c1, c2 = _u32(c1), _u32(c2)
        if c1.shape != c2.shape:
            raise ValueError("Cannot add Ciphertexts because they are different sizes")
        out = np.empty_like(c1)
        lib().gswo_ct_add(self.ring.ref, _p(c1), _p(c2), _p(out), C.c_size_t(c1.size))
        return out

    def mult_const(self, c, constant: int) -> np.ndarray:
        """gsw_impl.rs:97-101 (in place there; returns the new matrix here)."""
        c = _u32(c).copy()
        lib().gswo_ct_mult_const(self.ring.ref, _p(c), C.c_uint32(constant), C.c_size_t(c.size))
        return c

    def mult(self, c1, c2) -> np.ndarray:
        """gsw_impl.rs:104-111: the plain N x N x N product."""
        c1, c2 = _u32(c1), _u32(c2)
        if c1.shape != c2.shape:
            raise ValueError("Cannot add Ciphertexts because they are different sizes")  # sic, gsw_impl.rs:105
        out = np.empty_like(c1)
        lib().gswo_ct_mult(self.ring.ref, _p(c1), _p(c2), _p(out), C.c_size_t(c1.shape[0]))
        return out

    def nand(self, c1, c2) -> np.ndarray:
        """gsw_impl.rs:114-123."""
        c1, c2 = _u32(c1), _u32(c2)
        out = np.empty_like(c1)
        lib().gswo_ct_nand(self.ring.ref, _p(c1), _p(c2), _p(out), C.c_size_t(c1.shape[0]))
        return out

    # ---- compact form ----
    def mul_gadget(self, c1c, c2c, flags: int = 0) -> np.ndarray:
        c1c, c2c = _u32(c1c), _u32(c2c)
        assert c1c.shape == c2c.shape == (self.big_n, self.n + 1)
        out = np.empty_like(c1c)
        lib().gswo_mul_gadget(self.ring.ref, C.c_size_t(self.n), _p(c1c), _p(c2c), _p(out), C.c_uint(flags))
        return out

    def mul_gadget_rows(self, c1c, c2c, row0: int, nrows: int, flags: int = 0) -> np.ndarray:
        c1c, c2c = _u32(c1c), _u32(c2c)
        out = np.empty((nrows, self.n + 1), dtype=np.uint32)
        lib().gswo_mul_gadget_rows(self.ring.ref, C.c_size_t(self.n), _p(c1c), _p(c2c), _p(out), C.c_size_t(row0),
                                   C.c_size_t(nrows), C.c_uint(flags))
        return out

    def mul_gadget_batch(self, c1c, c2c, flags: int = 0, threads: int = 0) -> np.ndarray:
        c1c, c2c = _u32(c1c), _u32(c2c)
        batch = c1c.shape[0]
        per = self.big_n * (self.n + 1)
        out = np.empty_like(c1c)
        lib().gswo_mul_gadget_batch(self.ring.ref, C.c_size_t(self.n), _p(c1c), C.c_size_t(per), _p(c2c),
                                    C.c_size_t(per), _p(out), C.c_size_t(per), C.c_size_t(batch), C.c_uint(flags),
                                    C.c_int(threads))
        return out

    def decrypt_compact(self, sk: SecretKey, cc) -> int:
        cc = _u32(cc)
        return int(lib().gswo_decrypt_compact(self.ring.ref, C.c_size_t(self.n), _p(_u32(sk.s)), _p(cc)))

    def mp_decrypt_compact(self, sk: SecretKey, cc) -> int:
        cc = _u32(cc)
        out = C.c_uint32(0)
        rc = lib().gswo_mp_decrypt_compact(self.ring.ref, C.c_size_t(self.n), _p(_u32(sk.s)), _p(cc), C.byref(out))
        if rc != 0:
            raise RuntimeError("Only supported for pow2 rings!")
        return int(out.value)

    def row_dot_s(self, sk: SecretKey, cc, row: int) -> int:
        cc = _u32(cc)
        return int(lib().gswo_compact_row_dot_s(self.ring.ref, C.c_size_t(self.n), _p(_u32(sk.s)), _p(cc),
                                                C.c_size_t(row)))

    def noise(self, sk: SecretKey, cc, message: int) -> int:
        """Centered |<Cc[ell-1,:], s> - message * 2^(ell-1)| (the quantity the window test bounds)."""
        r = self.ring
        x = self.row_dot_s(sk, cc, r.ell - 1)
        e = r.sub(x, r.mul(message, r.from_u64(1 << (r.ell - 1))))
        return min(e, r.q - e)
