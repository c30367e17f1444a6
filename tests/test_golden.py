"""The C oracle against the committed golden vectors (tests/golden/*.npz), which were produced by the
independent big-integer restatement oracle/pyref.py (tests/golden/make_golden.py).

Provenance: the Rust reference cannot be compiled here and holds no vectors for mult/nand/add/
mult_const ("parity unpinned"); these vectors pin two separately written restatements of
gsw_impl.rs / helper.rs / sk.rs / pk.rs against each other.
"""
import glob
import os

import numpy as np
import pytest

GOLDEN = sorted(glob.glob(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "*.npz")))


def load(path):
    return dict(np.load(path, allow_pickle=False))


def ring_of(oracle, g):
    q, ell = int(g["q"]), int(g["ell"])
    return oracle.fp() if q == 2147483647 else oracle.zpow2(ell)


@pytest.mark.parametrize("path", GOLDEN, ids=[os.path.basename(p)[:-4] for p in GOLDEN])
def test_oracle_matches_golden(oracle, path):
    g = load(path)
    r = ring_of(oracle, g)
    n, m = int(g["n"]), int(g["m"])
    gsw = oracle.GSW(r, n, m)
    sk = oracle.sk_new(r, g["t"])
    assert np.array_equal(sk.s, g["s"]) and np.array_equal(sk.v, g["v"])  # sk.rs:20-28
    pk = oracle.pk_new(r, g["rand_b"], g["e"], g["t"])
    assert np.array_equal(pk.b, g["b"]) and np.array_equal(pk.pk_matrix, g["pk_matrix"])  # pk.rs:23-39
    # key invariant pk_matrix * s == e (gsw_impl.rs:150-163)
    assert np.array_equal(oracle.matmul(r, pk.pk_matrix, sk.s.reshape(-1, 1)).reshape(-1), g["e"])
    msgs = g["messages"].tolist()
    for i, mu in enumerate(msgs):
        assert np.array_equal(gsw.encrypt_compact(pk, mu, g["rbits"][i]), g["compact"][i])
        if mu in (0, 1):
            assert gsw.decrypt_compact(sk, g["compact"][i]) == int(g["decrypt_compact"][i]) == mu
    c1, c2 = g["compact"][0], g["compact"][1]
    assert np.array_equal(gsw.mul_gadget(c1, c2), g["mul_gadget"])
    assert np.array_equal(gsw.mul_gadget(c1, c2, flags=oracle.FLAG_NAND), g["nand_gadget"])
    assert np.array_equal(oracle.bit_decomp_matrix(r, g["mul_gadget"]), g["mul_gadget_flat"])
    assert np.array_equal(gsw.add(c1, c2), g["add_compact"])
    assert np.array_equal(gsw.mult_const(c1, 12345 % r.q), g["mult_const_compact"])
    if msgs[0] in (0, 1) and msgs[1] in (0, 1):
        assert gsw.decrypt_compact(sk, g["mul_gadget"]) == int(g["decrypt_mul"][0]) == msgs[0] * msgs[1]
        assert gsw.decrypt_compact(sk, g["nand_gadget"]) == 1 - msgs[0] * msgs[1]
    if "dense" in g:
        for i, mu in enumerate(msgs):
            assert np.array_equal(gsw.encrypt(pk, mu, g["rbits"][i]), g["dense"][i])  # gsw_impl.rs:27-42
            assert gsw.decrypt(sk, g["dense"][i]) == int(g["decrypt_dense"][i])
        d1, d2 = g["dense"][0], g["dense"][1]
        assert np.array_equal(gsw.mult(d1, d2), g["mult_dense_ref_format"])  # gsw_impl.rs:104-111
        assert np.array_equal(gsw.nand(d1, d2), g["nand_dense_ref_format"])  # gsw_impl.rs:114-123
        assert np.array_equal(gsw.add(d1, d2), g["add_dense_ref_format"])    # gsw_impl.rs:87-94
        f1, f2 = oracle.bit_decomp_matrix(r, c1), oracle.bit_decomp_matrix(r, c2)
        assert np.array_equal(gsw.mult(f1, f2), g["mult_dense_flat"])
        # the identity the throughput path rests on
        assert np.array_equal(oracle.bit_decomp_inv_matrix(r, g["mult_dense_flat"], quirk=False), g["mul_gadget"])
        assert np.array_equal(oracle.flatten_matrix(r, g["mult_dense_ref_format"]), g["flatten_quirk_of_product"])
        if "mp_decrypt_dense" in g:
            for i in range(len(msgs)):
                assert gsw.mp_decrypt(sk, g["dense"][i]) == int(g["mp_decrypt_dense"][i])  # gsw_impl.rs:67-84
                assert gsw.mp_decrypt_compact(sk, g["compact"][i]) == int(g["mp_decrypt_dense"][i])


def test_full_width_messages_round_trip():
    """gsw_impl.rs:189-207: Zpow2<31> full-width messages come back from mp_decrypt."""
    g = load([p for p in GOLDEN if "z31" in p][0])
    assert g["mp_decrypt_dense"].tolist() == g["messages"].tolist()
