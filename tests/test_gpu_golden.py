"""The HIP engine (through the C ABI) against the committed golden vectors of tests/golden/."""
import glob
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

GOLDEN = sorted(glob.glob(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "*.npz")))


@pytest.mark.parametrize("path", GOLDEN, ids=[os.path.basename(p)[:-4] for p in GOLDEN])
def test_engine_matches_golden(path):
    import tfhe_amd

    g = dict(np.load(path, allow_pickle=False))
    q, ell, n, m = int(g["q"]), int(g["ell"]), int(g["n"]), int(g["m"])
    ring = tfhe_amd.Fp() if q == 2147483647 else tfhe_amd.Zpow2(ell)
    gsw = tfhe_amd.GSW(ring, n, m, check_max=False)
    sk = gsw.make_sk(g["t"])
    assert np.array_equal(sk.s, g["s"]) and np.array_equal(sk.v, g["v"])
    pk = gsw.make_pk(g["rand_b"], g["e"], g["t"])
    assert np.array_equal(pk.b, g["b"]) and np.array_equal(pk.pk_matrix, g["pk_matrix"])
    msgs = g["messages"].tolist()
    ct = gsw.encrypt_batch(pk, msgs, rbits=g["rbits"])
    assert np.array_equal(ct.to_compact(), g["compact"])
    binary = all(mu in (0, 1) for mu in msgs)
    if binary:
        assert gsw.decrypt_batch(sk, ct).tolist() == g["decrypt_compact"].tolist() == msgs
    c1 = tfhe_amd.Ciphertext.from_compact(gsw.engine, g["compact"][0])
    c2 = tfhe_amd.Ciphertext.from_compact(gsw.engine, g["compact"][1])
    prod, nand = gsw.mult(c1, c2), gsw.nand(c1, c2)
    assert np.array_equal(prod.to_compact(), g["mul_gadget"])
    assert np.array_equal(nand.to_compact(), g["nand_gadget"])
    assert np.array_equal(prod.to_dense(), g["mul_gadget_flat"])  # G^-1 of the product
    assert np.array_equal(gsw.add(c1, c2).to_compact(), g["add_compact"])
    gsw.mult_const(c1, 12345 % q)
    assert np.array_equal(c1.to_compact(), g["mult_const_compact"])
    if binary:
        assert gsw.decrypt(sk, prod) == int(g["decrypt_mul"][0])
        assert gsw.decrypt(sk, nand) == 1 - int(g["decrypt_mul"][0])
    if "dense" in g:
        d1, d2 = g["dense"][0], g["dense"][1]
        # the reference's own ciphertext format as the left operand: the trait's mult / nand on the MFMA path ...
        gsw.set_dense_fast(2)  # at any size (by default only from 1024 rows up, where it pays)
        assert np.array_equal(gsw.mult_dense(d1, d2), g["mult_dense_ref_format"]) and gsw.last_dense_path() == 1
        assert np.array_equal(gsw.nand_dense(d1, d2), g["nand_dense_ref_format"]) and gsw.last_dense_path() == 1
        # ... and on the plain N^3 kernel
        gsw.set_dense_fast(False)
        assert np.array_equal(gsw.mult_dense(d1, d2), g["mult_dense_ref_format"]) and gsw.last_dense_path() == 0
        assert np.array_equal(gsw.nand_dense(d1, d2), g["nand_dense_ref_format"])
        gsw.set_dense_fast(1)
        assert np.array_equal(gsw.add_dense(d1, d2), g["add_dense_ref_format"])
        assert np.array_equal(gsw.flatten_matrix(g["mult_dense_ref_format"]), g["flatten_quirk_of_product"])
        assert [gsw.decrypt_dense(sk, d) for d in g["dense"]] == g["decrypt_dense"].tolist()
        assert np.array_equal(gsw.bit_decomp_inv_matrix(g["mult_dense_flat"], quirk=False), g["mul_gadget"])
    if "mp_decrypt_dense" in g:
        assert gsw.mp_decrypt_batch(sk, ct).tolist() == g["mp_decrypt_dense"].tolist()


def test_engine_on_the_reference_held_products():
    """The matrix products written out in the reference's own test code (helper.rs:166-183, field.rs:97-122,
    pow2_ring.rs:231-260; tests/golden/reference_held_products.json) through the C ABI: square cases through
    gsw_mul_dense / gsw_nand_dense (the trait's mult / nand, gsw_impl.rs:104-123), matrix-vector cases through
    gsw_matvec."""
    import json

    import tfhe_amd
    from tfhe_amd import _lib

    with open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "reference_held_products.json")) as f:
        cases = json.load(f)["cases"]
    L = _lib.lib()
    for case in cases:
        ring = tfhe_amd.Fp() if case["ring"] == "fp" else tfhe_amd.Zpow2(int(case["ring"][1:]))
        gsw = tfhe_amd.GSW(ring, 1, ring.ell)  # (n, m) play no part in these products
        eng = gsw.engine
        a = np.array(case["a"], dtype=np.uint32)
        expect = np.array(case["expect"], dtype=np.uint32)
        if case["op"] == "matmul":
            b = np.array(case["b"], dtype=np.uint32)
            assert np.array_equal(gsw.mult_dense(a, b), expect), case["cite"]
            assert gsw.last_dense_path() == 0  # arbitrary entries, small: the plain dim^3 kernel
            want = (np.eye(a.shape[0], dtype=np.int64) - expect.astype(np.int64)) % ring.q
            assert np.array_equal(gsw.nand_dense(a, b), want.astype(np.uint32)), case["cite"]
            gsw.set_dense_fast(2)  # and on the digit x digit MFMA products
            assert np.array_equal(gsw.mult_dense(a, b), expect) and gsw.last_dense_path() == 2, case["cite"]
            assert np.array_equal(gsw.nand_dense(a, b), want.astype(np.uint32)), case["cite"]
        else:
            vec = np.array(case["b"], dtype=np.uint32)
            rows, cols = a.shape
            d_a, d_o = eng.alloc_u32(a.size), eng.alloc_u32(rows)
            d_a.upload(a)
            _lib.check(L.gsw_matvec(eng.handle, d_a.ptr, rows, cols, cols, vec.ctypes.data, d_o.ptr))
            assert d_o.download(np.uint32, rows).tolist() == expect.tolist(), case["cite"]
