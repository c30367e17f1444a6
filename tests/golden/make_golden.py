"""Generates the committed golden vectors with the independent big-integer restatement
(oracle/pyref.py, written against the Rust sources, not against gsw_oracle.c).

The Rust reference cannot be built in this environment and holds no vectors for mult/nand/add/
mult_const, so these files pin the two restatements (and the GPU engine) against each other, not
against outputs of the reference itself.  Run from the repository root:  python tests/golden/make_golden.py
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import pyref  # noqa: E402

HERE = os.path.dirname(os.path.abspath(__file__))


def u32(a):
    return pyref.to_u32(a)


def gaussian(rng, q, count):
    e = np.rint(rng.normal(0.0, 1.0, size=count)).astype(np.int64)
    return [int(x) % q for x in e]


def make_set(name, ring, n, seed, messages, dense=True):
    rng = np.random.default_rng(seed)
    q, ell = ring.q, ring.ell
    m = n * ell
    big_n = ell * (n + 1)
    t = [int(x) for x in rng.integers(0, q, size=n)]
    rand_b = [[int(x) for x in row] for row in rng.integers(0, q, size=(m, n))]
    e = gaussian(rng, q, m)
    _, s, v = pyref.sk_new(ring, t)
    b, pk = pyref.pk_new(ring, rand_b, e, t)
    out = dict(n=np.int64(n), m=np.int64(m), ell=np.int64(ell), q=np.uint64(q), t=u32(t), s=u32(s), v=u32(v),
               rand_b=u32(rand_b), e=u32(e), b=u32(b), pk_matrix=u32(pk), messages=u32(messages))
    rbits, compact, denses = [], [], []
    for mu in messages:
        r = rng.integers(0, 2, size=(big_n, m)).astype(np.uint8)
        prod = pyref.matmul(ring, r, pk)
        cc = prod.copy()
        for i in range(big_n):  # + mu * G, G[i][i div ell] = 2^(i mod ell)
            cc[i, i // ell] = (int(cc[i, i // ell]) + mu * ((1 << (i % ell)) % q)) % q
        rbits.append(r)
        compact.append(u32(cc))
        if dense:
            denses.append(u32(pyref.encrypt(ring, n, pk, r, mu)))
    out["rbits"] = np.stack(rbits)
    out["compact"] = np.stack(compact)
    c1c, c2c = compact[0], compact[1]
    flat1 = pyref.bit_decomp_matrix(ring, c1c)
    pc = pyref.matmul(ring, flat1, c2c)  # BitDecomp(C1c) * C2c
    gadget = np.zeros((big_n, n + 1), dtype=object)
    for i in range(big_n):
        gadget[i, i // ell] = (1 << (i % ell)) % q
    out["mul_gadget"] = u32(pc)
    out["nand_gadget"] = u32((gadget - pc) % q)
    out["mul_gadget_flat"] = u32(pyref.bit_decomp_matrix(ring, pc))
    out["add_compact"] = u32(pyref.add(ring, c1c, c2c))
    out["mult_const_compact"] = u32(pyref.mult_const(ring, c1c, 12345 % q))
    out["decrypt_compact"] = u32([ring.is_zero_one(int(np.dot(np.array(c[ell - 1], dtype=object), s)) % q)
                                  for c in compact])
    out["decrypt_mul"] = u32([ring.is_zero_one(int(np.dot(np.array(pc[ell - 1], dtype=object), s)) % q)])
    if dense:
        d1, d2 = denses[0], denses[1]
        flat2 = pyref.bit_decomp_matrix(ring, c2c)
        out["dense"] = np.stack(denses)
        out["decrypt_dense"] = u32([pyref.decrypt(ring, v, d) for d in denses])
        out["mult_dense_ref_format"] = u32(pyref.mult(ring, d1, d2))       # gsw_impl.rs:104-111 on its own ciphertexts
        out["nand_dense_ref_format"] = u32(pyref.nand(ring, d1, d2))
        out["add_dense_ref_format"] = u32(pyref.add(ring, d1, d2))
        out["mult_dense_flat"] = u32(pyref.mult(ring, flat1, flat2))       # same on the flattened forms
        out["flatten_quirk_of_product"] = u32(pyref.flatten_matrix(ring, out["mult_dense_ref_format"]))
        if ring.kind == "zpow2":
            out["mp_decrypt_dense"] = u32([pyref.mp_decrypt(ring, v, d) for d in denses])
    np.savez_compressed(os.path.join(HERE, name + ".npz"), **out)
    print(name, {k: getattr(val, "shape", None) for k, val in out.items() if hasattr(val, "shape") and val.ndim})


if __name__ == "__main__":
    make_set("fp_n2", pyref.Ring("fp"), 2, 11, [1, 1, 0])
    make_set("z32_n2", pyref.Ring("zpow2", 32), 2, 12, [1, 0, 1])
    make_set("z31_n2_full_width", pyref.Ring("zpow2", 31), 2, 13, [0x5A5A5A5A & 0x7FFFFFFF, 1, 0x12345678])
    make_set("fp_n10_default_params", pyref.Ring("fp"), 10, 14, [1, 1], dense=False)  # n=10, m=310 (gsw_impl.rs:152-153)
