"""The CPU oracle against every known-answer test and invariant the reference's own
unit tests hold (the only pins available: the Rust crate cannot be built here)."""
import numpy as np
import pytest

from oracle import pyref

P = 2147483647


# ---- field.rs:59-124 --------------------------------------------------------------
def test_fp_inverse(oracle):  # field.rs:67-78 inverse_test
    r = oracle.fp()
    assert r.add(1, r.neg(1)) == 0
    rnd = oracle.fill_uniform(r, 11, 0, 1, P - 1, 2000)
    for x in rnd.tolist():
        assert r.mul(x, oracle.fp_inv(x)) == 1


def test_fp_constants(oracle):  # field.rs:80-84 test_constants
    r = oracle.fp()
    assert r.max_u64() == P - 1
    assert r.ell == 31


def test_fp_from_u64_reduces(oracle):  # field.rs:86-94 + helper.rs:125 (Fp::from(random u64))
    r = oracle.fp()
    for x in [0, 1, P - 1, P, P + 1, 2**32, 2**62 + 12345, 2**64 - 1]:
        assert r.from_u64(x) == x % P


def test_fp_scalar_product(oracle):  # field.rs:97-103: (1,2,3).(4,5,6) = 32
    assert oracle.dot(oracle.fp(), [1, 2, 3], [4, 5, 6]) == 32


def test_fp_matrix_vector(oracle):  # field.rs:105-122: [[1,2,3],[4,5,6]] * (7,8,9) = (50,122)
    out = oracle.matmul(oracle.fp(), [[1, 2, 3], [4, 5, 6]], [[7], [8], [9]])
    assert out.reshape(-1).tolist() == [50, 122]


# ---- pow2_ring.rs:191-262 ---------------------------------------------------------
def test_z8_kats(oracle):
    z8 = oracle.zpow2(3)
    assert z8.add(3, 5) == 0  # pow2_ring.rs:200-204
    assert z8.sub(3, 5) == 6  # :207-211
    assert z8.mul(3, z8.from_u64(15)) == 5  # :214-218
    assert z8.neg(3) == 5  # :221-224
    assert z8.from_u64(24) == 0  # :227-230


def test_z8_nalgebra(oracle):
    z8 = oracle.zpow2(3)
    ident = np.eye(3, dtype=np.uint32)
    v = np.array([[1], [2], [3]], dtype=np.uint32)
    assert (oracle.matmul(z8, ident, v) == v).all()  # :233-238
    d = 2 * ident
    assert oracle.matmul(z8, d, [[3], [4], [5]]).reshape(-1).tolist() == [6, 0, 2]  # :241-250
    a = np.array([[1, 2, 3], [4, 5, 6], [0, 1, 2]], dtype=np.uint32)
    assert (oracle.matmul(z8, a, ident) == a).all()  # :253-261


# ---- every matrix product the reference's own test code writes out (tests/golden/reference_held_products.json) ----
def _reference_held_cases():
    import json
    import os

    with open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "reference_held_products.json")) as f:
        return json.load(f)["cases"]


def test_mult_on_the_reference_held_products(oracle):
    """The only pins of the operation `mult` is (`ciphertext1 * cipertext2`, gsw_impl.rs:107) that the reference's own
    files hold: helper.rs:166-183 ([[1,2],[3,4]] * [[5,6],[7,8]] = [[19,22],[43,50]] over Fp, commented out there),
    field.rs:97-122 and pow2_ring.rs:231-260.  Through the oracle's matmul, through its `mult` (square cases) and
    through the independent big-integer restatement."""
    from oracle import pyref

    for case in _reference_held_cases():
        r = oracle.fp() if case["ring"] == "fp" else oracle.zpow2(int(case["ring"][1:]))
        pr = pyref.Ring("fp") if case["ring"] == "fp" else pyref.Ring("zpow2", int(case["ring"][1:]))
        a = np.array(case["a"], dtype=np.uint32)
        b = np.array(case["b"], dtype=np.uint32)
        expect = np.array(case["expect"], dtype=np.uint32)
        if case["op"] == "matvec":
            b, expect = b.reshape(-1, 1), expect.reshape(-1, 1)
        assert np.array_equal(oracle.matmul(r, a, b), expect), case["cite"]
        assert np.array_equal(np.array(pyref.matmul(pr, a.tolist(), b.tolist()), dtype=np.uint32), expect), case["cite"]
        if case["op"] == "matmul":
            dim = a.shape[0]
            gsw = oracle.GSW(r, 1, r.ell)  # (n, m) play no part in mult: any square matrices of one size
            assert np.array_equal(gsw.mult(a, b), expect), case["cite"]
            # nand = I - C1 C2 (gsw_impl.rs:114-123) follows from the same product
            want = (np.eye(dim, dtype=np.int64) - expect.astype(np.int64)) % r.q
            assert np.array_equal(gsw.nand(a, b), want.astype(np.uint32)), case["cite"]


# ---- gsw/helper.rs:105-164 ----------------------------------------------------------
def test_bit_decomp_and_inv(oracle):  # helper.rs:121-144
    r = oracle.fp()
    rng = np.random.default_rng(5)
    for _ in range(10):
        raw = rng.integers(0, 2**64, size=10, dtype=np.uint64)
        inp = np.array([r.from_u64(int(x)) for x in raw], dtype=np.uint32).reshape(1, 10)
        dec = oracle.bit_decomp_matrix(r, inp)
        assert dec.shape == (1, 10 * 31)
        assert set(np.unique(dec).tolist()) <= {0, 1}
        assert (oracle.bit_decomp_inv_matrix(r, dec, quirk=True) == inp).all()
        assert (oracle.bit_decomp_inv_matrix(r, dec, quirk=False) == inp).all()
    m = np.full((2, 2), 2, dtype=np.uint32)
    dm = oracle.bit_decomp_matrix(r, m)
    assert not np.array_equal(dm, m)
    assert (oracle.bit_decomp_inv_matrix(r, dm) == m).all()


def test_bit_decomp_layout(oracle):  # helper.rs:7-13: little endian, element j at j*ell..
    r = oracle.fp()
    dec = oracle.bit_decomp_matrix(r, [[5, 2]])
    assert dec[0, :3].tolist() == [1, 0, 1] and dec[0, 31:33].tolist() == [0, 1]
    assert dec.sum() == 3


def test_bit_decomp_inv_quirk(oracle):  # helper.rs:31-35: only entries == 1 count
    r = oracle.fp()
    row = np.zeros((1, 31), dtype=np.uint32)
    row[0, 0], row[0, 1], row[0, 2], row[0, 3] = 1, 2, 5, 1
    assert oracle.bit_decomp_inv_matrix(r, row, quirk=True)[0, 0] == 0b1001
    assert oracle.bit_decomp_inv_matrix(r, row, quirk=False)[0, 0] == 1 + 4 + 20 + 8
    ones = np.ones((1, 31), dtype=np.uint32)  # repr = 2^31 - 1 = p -> Fp::from gives 0
    assert oracle.bit_decomp_inv_matrix(r, ones, quirk=True)[0, 0] == 0


def test_scalar_product_invariant(oracle):  # helper.rs:146-164
    for r in (oracle.fp(), oracle.zpow2(30), oracle.zpow2(32)):
        a = oracle.fill_uniform(r, 3, 1, 0, r.max_u64(), 10, checked=False)
        b = oracle.fill_uniform(r, 3, 2, 0, r.max_u64(), 10, checked=False)
        bd = oracle.bit_decomp_matrix(r, a.reshape(1, -1)).reshape(-1)
        po2 = oracle.powers_of_2(r, b)
        assert bd.size == po2.size
        assert oracle.dot(r, bd, po2) == oracle.dot(r, a, b)


# ---- gsw/sk.rs:33-45, gsw/pk.rs:43-65 -------------------------------------------------
def test_v_decomp(oracle):  # sk.rs:37-44
    r = oracle.fp()
    sk = oracle.sk_new(r, oracle.fill_uniform(r, 9, 0, 0, 10, 5))
    assert sk.v[:31].tolist() == [1 << i for i in range(31)]


def test_different_err_lead_to_diff_pk(oracle):  # pk.rs:49-64
    r = oracle.fp()
    n, m = 100, 50
    t = oracle.fill_uniform(r, 4, 0, 0, P - 1, n)
    b = oracle.fill_uniform(r, 4, 1, 0, 1, m * n).reshape(m, n)
    pk1 = oracle.pk_new(r, b, oracle.fill_uniform(r, 4, 2, 0, P // 2, m), t)
    pk2 = oracle.pk_new(r, b, oracle.fill_uniform(r, 4, 3, 0, P // 2, m), t)
    assert not np.array_equal(pk1.pk_matrix, pk2.pk_matrix)
    with pytest.raises(ValueError):
        oracle.pk_new(r, b, np.zeros(m + 1, dtype=np.uint32), t)


# ---- gsw/gsw_impl.rs:133-220 -----------------------------------------------------------
def test_sk_pk_invariant(oracle):  # gsw_impl.rs:150-163: pk_matrix * s == e exactly
    r = oracle.fp()
    n, m = 10, 310
    sk = oracle.sk_new(r, oracle.fill_uniform(r, 21, 0, 0, r.max_u64(), n))
    err = oracle.fill_uniform(r, 21, 1, 0, r.max_u64() >> 15, m)
    b = oracle.fill_uniform(r, 21, 2, 0, r.max_u64(), m * n).reshape(m, n)
    pk = oracle.pk_new(r, b, err, sk.t)
    assert (oracle.matmul(r, pk.pk_matrix, sk.s.reshape(-1, 1)).reshape(-1) == err).all()


def _enc_dec_bits(oracle, gsw):  # gsw_impl.rs:209-219 test_inputs
    sk, pk = gsw.keygen()
    for bit in (0, 1):
        assert gsw.decrypt(sk, gsw.encrypt(pk, bit)) == bit
        assert gsw.decrypt_compact(sk, gsw.encrypt_compact(pk, bit)) == bit


def test_encryption_decryption_naive(oracle):  # gsw_impl.rs:165-174
    _enc_dec_bits(oracle, oracle.GSW(oracle.fp(), 10, 310, sampler="naive", seed=7))


def test_encryption_decryption_discr_gaussian(oracle):  # gsw_impl.rs:177-186
    _enc_dec_bits(oracle, oracle.GSW(oracle.zpow2(30), 10, 300, sampler="gaussian", seed=8))


def test_encryption_decryption_pow_of_two(oracle):  # gsw_impl.rs:189-207
    r = oracle.zpow2(31)
    gsw = oracle.GSW(r, 10, 310, sampler="gaussian", seed=9)
    sk, pk = gsw.keygen()
    msgs = oracle.fill_uniform(r, 99, 0, 0, r.max_u64() - 1, 4)
    for msg in msgs.tolist():
        assert gsw.mp_decrypt(sk, gsw.encrypt(pk, msg)) == msg
        assert gsw.mp_decrypt_compact(sk, gsw.encrypt_compact(pk, msg)) == msg
    with pytest.raises(RuntimeError):  # gsw_impl.rs:68-70
        oracle.GSW(oracle.fp(), 2, 62).mp_decrypt(oracle.sk_new(oracle.fp(), [1, 2]), np.zeros((93, 93)))


def test_keygen_pow2_32_panics_like_reference(oracle):  # error_sampling.rs:21 via gsw_impl.rs:18
    with pytest.raises(AssertionError):
        oracle.GSW(oracle.zpow2(32), 4, 128).keygen()
    oracle.GSW(oracle.zpow2(32), 4, 128, check_max=False).keygen()


# ---- error_sampling.rs:86-105 and distribution checks -----------------------------------
def test_gaussian_sampler(oracle):
    r = oracle.fp()
    v = oracle.sample_gaussian(r, 5, 0, 1.0, 100)
    assert len(set(v.tolist())) > 1  # gaussian_test: not all equal
    s = oracle.sample_gaussian_i32(5, 1, 1.0, 200000).astype(np.float64)
    assert abs(s.mean()) < 0.02 and abs(s.var() - 1.0) < 0.03 and np.abs(s).max() <= 14
    s3 = oracle.sample_gaussian_i32(5, 2, 3.0, 200000).astype(np.float64)
    assert abs(s3.var() - 9.0) < 0.3
    neg = oracle.sample_gaussian(r, 5, 1, 1.0, 1000)
    assert ((neg < 16) | (neg > P - 16)).all()  # signed value mapped to +-R (error_sampling.rs:60-64)


def test_naive_sampler_range(oracle):  # error_sampling.rs:75-79
    v = oracle.sample_naive(oracle.fp(), 6, 0, 100000)
    assert v.max() <= (P >> 15) and v.max() > 60000 and v.min() < 1000


def test_uniform_asserts_max_le_p(oracle):  # error_sampling.rs:21
    with pytest.raises(AssertionError):
        oracle.fill_uniform(oracle.zpow2(32), 1, 0, 0, 2**32 - 1, 4)


# ---- the compact-form identity and the four unpinned operations ------------------------
@pytest.mark.parametrize("ring_name", ["fp", "z30", "z32"])
def test_compact_identity_and_ops(oracle, ring_name):
    """BitDecomp^-1_linear(C1 * C2) == BitDecomp(C1c) * C2c, and both C restatements agree
    with the independent big-int restatement (mult/nand/add/mult_const are parity unpinned:
    the reference holds no vectors for them)."""
    r = {"fp": oracle.fp(), "z30": oracle.zpow2(30), "z32": oracle.zpow2(32)}[ring_name]
    pr = pyref.Ring("fp") if ring_name == "fp" else pyref.Ring("zpow2", r.ell)
    n = 2
    m = n * r.ell
    gsw = oracle.GSW(r, n, m, sampler="gaussian", seed=31, check_max=False)
    sk, pk = gsw.keygen()
    for mu1, mu2 in ((1, 1), (1, 0), (0, 1)):
        c1c, c2c = gsw.encrypt_compact(pk, mu1), gsw.encrypt_compact(pk, mu2)
        d1, d2 = oracle.bit_decomp_matrix(r, c1c), oracle.bit_decomp_matrix(r, c2c)
        dense = gsw.mult(d1, d2)
        assert np.array_equal(pyref.to_u32(pyref.mult(pr, d1, d2)), dense)
        pc = gsw.mul_gadget(c1c, c2c)
        assert np.array_equal(oracle.bit_decomp_inv_matrix(r, dense, quirk=False), pc)
        assert gsw.decrypt_compact(sk, pc) == mu1 * mu2
        nd = gsw.nand(d1, d2)
        assert np.array_equal(pyref.to_u32(pyref.nand(pr, d1, d2)), nd)
        nc = gsw.mul_gadget(c1c, c2c, flags=oracle.FLAG_NAND)
        assert np.array_equal(oracle.bit_decomp_inv_matrix(r, nd, quirk=False), nc)
        assert gsw.decrypt_compact(sk, nc) == 1 - mu1 * mu2
        assert np.array_equal(pyref.to_u32(pyref.add(pr, d1, d2)), gsw.add(d1, d2))
        assert np.array_equal(pyref.to_u32(pyref.mult_const(pr, d1, 12345)), gsw.mult_const(d1, 12345))
        assert np.array_equal(pyref.to_u32(pyref.flatten_matrix(pr, dense)), oracle.flatten_matrix(r, dense))


def test_reference_format_ciphertext_mult_decrypts(oracle):
    """The reference's own dense ciphertext (BitDecomp(R*A) + mu*I, gsw_impl.rs:27-42) through
    the dense product of gsw_impl.rs:107, with sigma = 1 errors (NaiveSampler noise is too
    large for a product at these parameters)."""
    r = oracle.fp()
    gsw = oracle.GSW(r, 3, 93, sampler="gaussian", seed=17)
    pr = pyref.Ring("fp")
    sk, pk = gsw.keygen()
    for mu1, mu2 in ((1, 1), (0, 1), (1, 0), (0, 0)):
        rb1, rb2 = gsw.random_bits(), gsw.random_bits()
        c1, c2 = gsw.encrypt(pk, mu1, rb1), gsw.encrypt(pk, mu2, rb2)
        assert np.array_equal(pyref.to_u32(pyref.encrypt(pr, 3, pk.pk_matrix, rb1, mu1)), c1)
        assert gsw.decrypt(sk, c1) == mu1 == pyref.decrypt(pr, sk.v, c1)
        assert gsw.decrypt(sk, gsw.mult(c1, c2)) == mu1 * mu2
        assert gsw.decrypt(sk, gsw.nand(c1, c2)) == 1 - mu1 * mu2
