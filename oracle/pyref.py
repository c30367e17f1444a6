"""Independent big-integer restatement of the reference's GSW arithmetic.

TEST INFRASTRUCTURE ONLY.  Written against the Rust sources directly (not against
gsw_oracle.c) with Python's unbounded ints, so that two separately written restatements
must agree on the committed golden vectors.  Slow: use for n <= 10.

Matrices are lists of rows / numpy object arrays (row-major); elements are canonical ints.
"""
from __future__ import annotations

import numpy as np

P = 2147483647  # field.rs:19


class Ring:
    def __init__(self, kind: str, m: int = 31):
        assert kind in ("fp", "zpow2")
        self.kind = kind
        self.ell = 31 if kind == "fp" else m  # field.rs:39, pow2_ring.rs:19
        self.q = P if kind == "fp" else (1 << m)

    def frm(self, x: int) -> int:  # From<u64>
        return (x % (1 << 64)) % self.q

    def max_u64(self) -> int:  # field.rs:36-38,47-49; pow2_ring.rs:31-33
        return self.q - 1

    def is_zero_one(self, x: int) -> int:  # gsw_impl.rs:127-131
        mx = self.max_u64()
        return 1 if self.frm(mx // 4) <= x <= self.frm(3 * mx // 4) else 0


def _obj(a):
    return np.array(a, dtype=object)


def matmul(r: Ring, a, b):
    return (_obj(a).dot(_obj(b))) % r.q


def bit_decomp_matrix(r: Ring, a):  # helper.rs:7-21
    a = _obj(a)
    rows, cols = a.shape
    out = np.zeros((rows, cols * r.ell), dtype=object)
    for i in range(rows):
        for j in range(cols):
            x = int(a[i, j])
            for k in range(r.ell):
                out[i, j * r.ell + k] = (x >> k) & 1
    return out


def bit_decomp_inv_matrix(r: Ring, a, quirk=True):  # helper.rs:24-49
    a = _obj(a)
    rows, wide = a.shape
    cols = wide // r.ell
    out = np.zeros((rows, cols), dtype=object)
    for i in range(rows):
        for j in range(cols):
            if quirk:
                repr_ = 0
                for k in range(r.ell):
                    if int(a[i, j * r.ell + k]) == 1:
                        repr_ |= 1 << k
                out[i, j] = r.frm(repr_)
            else:
                out[i, j] = sum(int(a[i, j * r.ell + k]) << k for k in range(r.ell)) % r.q
    return out


def flatten_matrix(r: Ring, a):  # helper.rs:51-62
    return bit_decomp_matrix(r, bit_decomp_inv_matrix(r, a, quirk=True))


def powers_of_2(r: Ring, b):  # helper.rs:65-77 with gsw/mod.rs:58-62
    return _obj([(int(x) * r.frm(1 << k)) % r.q for x in b for k in range(r.ell)])


def sk_new(r: Ring, t):  # sk.rs:20-28
    s = [1 % r.q] + [(-int(x)) % r.q for x in t]
    return _obj(list(t)), _obj(s), powers_of_2(r, s)


def pk_new(r: Ring, rand_b, e, t):  # pk.rs:23-39
    rand_b, e, t = _obj(rand_b), _obj(e), _obj(t)
    b = (rand_b.dot(t) + e) % r.q
    return b, np.concatenate([b.reshape(-1, 1), rand_b], axis=1)


def encrypt(r: Ring, n: int, pk_matrix, rbits, message: int):  # gsw_impl.rs:27-42
    prod = matmul(r, _obj(rbits), pk_matrix)
    c = bit_decomp_matrix(r, prod)
    for i in range(c.shape[1]):
        c[i, i] = (int(c[i, i]) + message) % r.q
    return c


def decrypt(r: Ring, v, c) -> int:  # gsw_impl.rs:48-52
    i = r.ell - 1
    x = int(_obj(c)[i].dot(_obj(v))) % r.q
    return r.is_zero_one(x)


def mp_decrypt(r: Ring, v, c) -> int:  # gsw_impl.rs:67-84
    if r.kind == "fp":
        raise RuntimeError("Only supported for pow2 rings!")
    product = matmul(r, c, _obj(v).reshape(-1, 1)).reshape(-1)
    slice_ = [int(x) for x in product[: r.ell]]
    recovered = 0
    for i, entry in enumerate(reversed(slice_)):
        recovered_exp = r.frm(recovered << (r.ell - i - 1))
        current = (entry - recovered_exp) % r.q
        recovered ^= r.is_zero_one(current) << i
    return r.frm(recovered)


def add(r: Ring, c1, c2):  # gsw_impl.rs:87-94
    return (_obj(c1) + _obj(c2)) % r.q


def mult_const(r: Ring, c, k: int):  # gsw_impl.rs:97-101
    return (_obj(c) * int(k)) % r.q


def mult(r: Ring, c1, c2):  # gsw_impl.rs:104-111
    return matmul(r, c1, c2)


def nand(r: Ring, c1, c2):  # gsw_impl.rs:114-123
    prod = (-matmul(r, c1, c2)) % r.q
    for i in range(prod.shape[1]):
        prod[i, i] = (int(prod[i, i]) + 1) % r.q
    return prod


def to_u32(a) -> np.ndarray:
    return np.array(a, dtype=np.uint64).astype(np.uint32)
