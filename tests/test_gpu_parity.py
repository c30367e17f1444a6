"""GPU parity tests: the HIP engine, called through the C ABI, against the CPU oracle on the
same inputs -- bit exact (all arithmetic is integer) -- plus decrypt-after-operation.

The reference holds no vectors for mult/nand/add/mult_const ("parity unpinned"): they are
checked against the oracle's restatement of gsw_impl.rs:87-123 and by decryption.
"""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

P = 2147483647


def _rings(oracle, tfhe_amd):
    return {
        "fp": (oracle.fp(), tfhe_amd.Fp()),
        "z30": (oracle.zpow2(30), tfhe_amd.Zpow2(30)),
        "z31": (oracle.zpow2(31), tfhe_amd.Zpow2(31)),
        "z32": (oracle.zpow2(32), tfhe_amd.Zpow2(32)),
        "z8": (oracle.zpow2(8), tfhe_amd.Zpow2(8)),
        "z2": (oracle.zpow2(2), tfhe_amd.Zpow2(2)),
        "z1": (oracle.zpow2(1), tfhe_amd.Zpow2(1)),
    }


@pytest.fixture(scope="module")
def tfhe_amd():
    import tfhe_amd as pkg
    from tfhe_amd import _lib

    _lib.lib()  # raises if libgsw_hip.so is missing: there is no fallback
    return pkg


def _rand_compact(rng, ring_o, n, batch=None):
    shape = (ring_o.big_n(n), n + 1) if batch is None else (batch, ring_o.big_n(n), n + 1)
    return rng.integers(0, ring_o.q, size=shape, dtype=np.uint64).astype(np.uint32)


def _assert_equal(got, expect, what):
    if not np.array_equal(got, expect):
        bad = np.argwhere(got != expect)
        raise AssertionError(f"{what}: {len(bad)} of {got.size} entries differ; first at {bad[0].tolist()}: "
                             f"got {got[tuple(bad[0])]}, expect {expect[tuple(bad[0])]}")


# ---------------------------------------------------------------- the hot path, small shapes
# kernel variants of the shipped library: 9 = the default (16x16x64 MFMA, skewed pipeline, packed tail tile),
# 2 / 3 = the simple two-buffer 32x32x32 kernels (LDS-DMA / register staging), an independent cross-check.
# n + 1 columns: 2, 4 (a narrow tile only), 8 (a packed tile only), 11, 33/34/35/36 (full + narrow tail), 37/40 (full +
# packed tail), 41 (full + unpacked tail), 67 (two full + narrow tail), 72 (two full + packed tail of 8)
VARIANTS = ["2", "3", "9"]


@pytest.mark.parametrize("variant", VARIANTS)
@pytest.mark.parametrize("ring_name,n", [("fp", 10), ("fp", 1), ("fp", 3), ("fp", 7), ("fp", 32), ("fp", 39),
                                          ("fp", 40), ("fp", 71), ("z30", 10), ("z31", 10), ("z32", 10),
                                          ("z32", 33), ("z8", 5), ("z2", 9), ("z1", 40),
                                          # last row tile of at most 64 rows (the thin-tile path of variant 9)
                                          ("fp", 8), ("fp", 16), ("z32", 8),
                                          # tail tiles of 2, 3, 4 columns (stored narrow: 16 MFMA columns) and of 5 (packed: 32)
                                          ("fp", 33), ("fp", 34), ("z32", 35), ("fp", 36), ("z31", 66)])
def test_mul_gadget_random_operands(oracle, tfhe_amd, ring_name, n, variant):
    """Pc = BitDecomp(C1c) * C2c on uniform random operands (every bit pattern exercised)."""
    ring_o, ring_d = _rings(oracle, tfhe_amd)[ring_name]
    os.environ["GSW_GEMM_VARIANT"] = variant
    try:
        gsw = tfhe_amd.GSW(ring_d, n, n * ring_d.ell, seed=1)
    finally:
        os.environ.pop("GSW_GEMM_VARIANT", None)
    og = oracle.GSW(ring_o, n, n * ring_o.ell)
    rng = np.random.default_rng(100 + n)
    c1, c2 = _rand_compact(rng, ring_o, n, 3), _rand_compact(rng, ring_o, n, 3)
    c1[2] = ring_o.q - 1  # all-ones rows: the largest accumulators the shape allows
    c2[2] = ring_o.q - 1
    d1 = tfhe_amd.Ciphertext.from_compact(gsw.engine, c1)
    d2 = tfhe_amd.Ciphertext.from_compact(gsw.engine, c2)
    _assert_equal(d1.to_compact(), c1, "upload/download round trip")
    got = gsw.mult(d1, d2).to_compact()
    got_nand = gsw.nand(d1, d2).to_compact()
    for b in range(3):
        _assert_equal(got[b], og.mul_gadget(c1[b], c2[b]), f"mult[{b}] {ring_name} n={n}")
        _assert_equal(got_nand[b], og.mul_gadget(c1[b], c2[b], flags=oracle.FLAG_NAND), f"nand[{b}] {ring_name} n={n}")


def test_mul_gadget_padding_stays_zero_and_strides(oracle, tfhe_amd):
    """Padding columns of the device layout stay zero; stride 0 broadcasts one right operand."""
    ring_o, ring_d = _rings(oracle, tfhe_amd)["fp"]
    n = 10
    gsw = tfhe_amd.GSW(ring_d, n, n * 31)
    og = oracle.GSW(ring_o, n, n * 31)
    eng = gsw.engine
    from tfhe_amd import _lib
    rng = np.random.default_rng(7)
    c1, c2 = _rand_compact(rng, ring_o, n, 4), _rand_compact(rng, ring_o, n)
    d1 = tfhe_amd.Ciphertext.from_compact(eng, c1)
    d2 = tfhe_amd.Ciphertext.from_compact(eng, c2)
    out = tfhe_amd.Ciphertext(eng, 4)
    _lib.check(_lib.lib().gsw_mul_gadget_batch(eng.handle, d1.ptr, d1.stride, d2.ptr, 0, out.ptr, out.stride, 4, 0))
    got = out.to_compact()
    for b in range(4):
        _assert_equal(got[b], og.mul_gadget(c1[b], c2), f"broadcast rhs [{b}]")
    raw = out.buf.download(np.uint32, eng.compact_elems * 4).reshape(4, eng.big_n, eng.ld)
    assert eng.ld == 16 and (raw[:, :, n + 1:] == 0).all()
    # digests from the contraction's epilogue == the digest kernel == the oracle's digest of the buffer
    dig = eng.alloc(8 * 4)
    dig2 = eng.alloc(8 * 4)
    _lib.check(_lib.lib().gsw_mul_gadget_batch_digest(eng.handle, d1.ptr, d1.stride, d2.ptr, 0, out.ptr, out.stride,
                                                      4, _lib.FLAG_NAND, dig.ptr))
    _lib.check(_lib.lib().gsw_digest_batch(eng.handle, out.ptr, out.stride, eng.compact_elems, 4, dig2.ptr))
    raw = out.buf.download(np.uint32, eng.compact_elems * 4).reshape(4, -1)
    want = [oracle.digest(raw[b]) for b in range(4)]
    assert dig.download(np.uint64, 4).tolist() == want == dig2.download(np.uint64, 4).tolist()
    for b in range(4):
        _assert_equal(out.to_compact()[b], og.mul_gadget(c1[b], c2, flags=oracle.FLAG_NAND), f"nand+digest [{b}]")
    # empty batch is a no-op
    _lib.check(_lib.lib().gsw_mul_gadget_batch(eng.handle, d1.ptr, d1.stride, d2.ptr, 0, out.ptr, out.stride, 0, 0))
    # an output that shares memory with an operand is refused (tiles are written while others still read)
    for dst, count in ((d1.ptr, 1), (d1.entry_ptr(3), 4), (d2.ptr, 1)):
        with pytest.raises(tfhe_amd.GswError, match="overlaps"):
            _lib.check(_lib.lib().gsw_mul_gadget_batch(eng.handle, d1.ptr, d1.stride, d2.ptr, 0, dst, out.stride,
                                                       count, 0))
    with pytest.raises(tfhe_amd.GswError):
        _lib.check(_lib.lib().gsw_mul_gadget_batch(eng.handle, d1.ptr + 4, d1.stride, d2.ptr, 0, out.ptr, out.stride,
                                                   1, 0))


@pytest.mark.parametrize("variant", VARIANTS)
@pytest.mark.parametrize("n,align", [(32, "16"), (32, "32"), (39, "16"), (64, "32"), (7, "32")])
def test_padding_columns_are_written_even_into_a_dirty_buffer(oracle, tfhe_amd, n, align, variant):
    """The output buffer starts as garbage: every padding column up to the row pitch must come out zero (a
    packed tail tile computes 8 columns but the pitch can reach 32 past its first), and the digest taken in
    the epilogue must equal the digest of the stored buffer."""
    from tfhe_amd import _lib

    L = _lib.lib()
    ring_o, ring_d = _rings(oracle, tfhe_amd)["fp"]
    os.environ["GSW_GEMM_VARIANT"] = variant
    os.environ["GSW_LD_ALIGN"] = align
    try:
        gsw = tfhe_amd.GSW(ring_d, n, n * 31, seed=1)
    finally:
        os.environ.pop("GSW_GEMM_VARIANT", None)
        os.environ.pop("GSW_LD_ALIGN", None)
    eng = gsw.engine
    og = oracle.GSW(ring_o, n, n * 31)
    rng = np.random.default_rng(n)
    c1, c2 = _rand_compact(rng, ring_o, n, 2), _rand_compact(rng, ring_o, n, 2)
    d1, d2 = tfhe_amd.Ciphertext.from_compact(eng, c1), tfhe_amd.Ciphertext.from_compact(eng, c2)
    out = tfhe_amd.Ciphertext(eng, 2)
    out.buf.upload(np.full(2 * eng.compact_elems, 0xDEADBEEF, dtype=np.uint32))
    dig, dig2 = eng.alloc(16), eng.alloc(16)
    _lib.check(L.gsw_mul_gadget_batch_digest(eng.handle, d1.ptr, d1.stride, d2.ptr, d2.stride, out.ptr, out.stride, 2,
                                             0, dig.ptr))
    raw = out.buf.download(np.uint32, 2 * eng.compact_elems).reshape(2, eng.big_n, eng.ld)
    assert (raw[:, :, n + 1:] == 0).all(), np.argwhere(raw[:, :, n + 1:] != 0)[:4]
    for b in range(2):
        _assert_equal(raw[b, :, :n + 1], og.mul_gadget(c1[b], c2[b]), f"product [{b}]")
    _lib.check(L.gsw_digest_batch(eng.handle, out.ptr, out.stride, eng.compact_elems, 2, dig2.ptr))
    assert dig.download(np.uint64, 2).tolist() == dig2.download(np.uint64, 2).tolist()


def test_bit_decomp_inv_writes_the_padding_columns(oracle, tfhe_amd):
    """Dense -> compact conversion into a dirty buffer: padding columns come out zero."""
    from tfhe_amd import _lib

    L = _lib.lib()
    ring_o, ring_d = _rings(oracle, tfhe_amd)["fp"]
    n = 4
    gsw = tfhe_amd.GSW(ring_d, n, n * 31, seed=1)
    eng = gsw.engine
    rng = np.random.default_rng(2)
    cc = _rand_compact(rng, ring_o, n)
    dense = oracle.bit_decomp_matrix(ring_o, cc)
    d_dense = eng.alloc_u32(dense.size)
    d_dense.upload(np.ascontiguousarray(dense, dtype=np.uint32))
    out = tfhe_amd.Ciphertext(eng, 1)
    out.buf.upload(np.full(eng.compact_elems, 0xDEADBEEF, dtype=np.uint32))
    _lib.check(L.gsw_bit_decomp_inv(eng.handle, d_dense.ptr, eng.big_n, n + 1, out.ptr, eng.ld, 0))
    raw = out.buf.download(np.uint32, eng.compact_elems).reshape(eng.big_n, eng.ld)
    _assert_equal(raw[:, :n + 1], cc, "BitDecomp^-1(BitDecomp(Cc))")
    assert eng.ld > n + 1 and (raw[:, n + 1:] == 0).all()


def test_empty_batches_and_ragged_public_keys(oracle, tfhe_amd):
    """batch = 0 is a no-op everywhere; m need not be a multiple of ell (the last word of a row of
    encryption bits is partly used) nor equal n * ell."""
    from tfhe_amd import _lib

    L = _lib.lib()
    ring_o, ring_d = _rings(oracle, tfhe_amd)["fp"]
    n, m = 5, 5 * 31 + 17
    gsw = tfhe_amd.GSW(ring_d, n, m, seed=8)
    eng = gsw.engine
    ct = tfhe_amd.Ciphertext(eng, 1)
    _lib.check(L.gsw_mul_gadget_batch(eng.handle, ct.ptr, ct.stride, ct.ptr, ct.stride, ct.ptr, ct.stride, 0, 0))
    _lib.check(L.gsw_prepare_rhs(eng.handle, ct.ptr, ct.stride, ct.ptr, 0))
    gsw.eval_circuit(ct, [])
    sk, pk = gsw.keygen()
    assert gsw.decrypt_batch(sk, ct, count=0).size == 0
    assert pk.pk_matrix.shape == (m, n + 1)
    og = oracle.GSW(ring_o, n, m)
    rng = np.random.default_rng(3)
    rbits = rng.integers(0, 2, size=(2, ring_o.big_n(n), m), dtype=np.uint8)
    got = gsw.encrypt_batch(pk, [1, 0], rbits=rbits)
    opk = oracle.PublicKey(pk.b, pk.pk_matrix)
    for b, msg in enumerate([1, 0]):
        _assert_equal(got.to_compact()[b], og.encrypt_compact(opk, msg, rbits[b]), f"ragged m, message {msg}")
    assert gsw.decrypt_batch(sk, got).tolist() == [1, 0]
    assert gsw.decrypt_batch(sk, gsw.encrypt_batch(pk, [0, 1, 1])).tolist() == [0, 1, 1]  # device-drawn bits


def test_workspace_chunking(oracle, tfhe_amd):
    """A batch larger than the workspace allows is processed in chunks with identical results."""
    ring_o, ring_d = _rings(oracle, tfhe_amd)["fp"]
    n = 5
    gsw = tfhe_amd.GSW(ring_d, n, n * 31)
    og = oracle.GSW(ring_o, n, n * 31)
    from tfhe_amd import _lib
    _lib.check(_lib.lib().gsw_ctx_set_workspace_limit(gsw.engine.handle, 2 * gsw.engine.prepared_bytes))
    rng = np.random.default_rng(8)
    c1, c2 = _rand_compact(rng, ring_o, n, 5), _rand_compact(rng, ring_o, n, 5)
    got = gsw.mult(tfhe_amd.Ciphertext.from_compact(gsw.engine, c1),
                   tfhe_amd.Ciphertext.from_compact(gsw.engine, c2)).to_compact()
    _assert_equal(got, og.mul_gadget_batch(c1, c2), "chunked batch")


# ---------------------------------------------------------------- scheme round trips
@pytest.mark.parametrize("ring_name,n,sampler", [("fp", 10, "gaussian"), ("z30", 10, "gaussian"),
                                                  ("z32", 6, "gaussian"), ("fp", 10, "naive")])
def test_keygen_encrypt_decrypt(oracle, tfhe_amd, ring_name, n, sampler):
    """gsw_impl.rs:165-186 (test_inputs) on the GPU engine, plus the key invariant of :150-163."""
    ring_o, ring_d = _rings(oracle, tfhe_amd)[ring_name]
    m = n * ring_d.ell
    gsw = tfhe_amd.GSW(ring_d, n, m, sampler=sampler, seed=42, check_max=(ring_name != "z32"))
    sk, pk = gsw.keygen()
    # pk_matrix * s == e: recompute e with the oracle and check it is a sample of the error distribution
    e = oracle.matmul(ring_o, pk.pk_matrix, sk.s.reshape(-1, 1)).reshape(-1).astype(np.int64)
    if sampler == "naive":
        assert e.max() <= (P >> 15)
    else:
        centered = np.where(e > ring_o.q // 2, e - ring_o.q, e)
        assert np.abs(centered).max() <= 14 and centered.std() > 0.5
    for bit in (0, 1):
        assert gsw.decrypt(sk, gsw.encrypt(pk, bit)) == bit
    ct = gsw.encrypt_batch(pk, [1, 0, 1, 1, 0])
    assert gsw.decrypt_batch(sk, ct).tolist() == [1, 0, 1, 1, 0]
    # the reference-format dense matrix of the same ciphertext decrypts the same (gsw_impl.rs:48-52)
    assert gsw.decrypt_dense(sk, ct.to_dense(0)) == 1
    assert gsw.decrypt_dense(sk, ct.to_dense(1)) == 0


@pytest.mark.parametrize("ring_name,n,sampler", [("fp", 10, "gaussian"), ("fp", 6, "naive"), ("z32", 5, "gaussian"),
                                                  ("z30", 10, 3.2), ("fp", 40, "gaussian")])
def test_device_keygen_equals_the_oracle_and_the_host_assembled_key(oracle, tfhe_amd, ring_name, n, sampler):
    """keygen on the device (samples, b = B t + e, pk_matrix, prepared form: nothing but t crosses to the host)
    gives, draw for draw, the oracle's keygen (gsw_impl.rs:17-25) of the same seed -- uniform draws bit for bit;
    the Gaussian errors come from the same table sampler on the same stream -- and the key that make_pk assembles
    through the host from the same draws; the prepared form encrypts like the oracle's pk."""
    ring_o, ring_d = _rings(oracle, tfhe_amd)[ring_name]
    m = n * ring_d.ell + 3
    checked = ring_name != "z32"
    gsw = tfhe_amd.GSW(ring_d, n, m, sampler=sampler, seed=99, check_max=checked)
    sk, pk = gsw.keygen()
    again = tfhe_amd.GSW(ring_d, n, m, sampler=sampler, seed=99, check_max=checked)
    sk2, pk2 = again.keygen_host_assembled()
    _assert_equal(sk.t, sk2.t, "t")
    _assert_equal(pk.pk_matrix, pk2.pk_matrix, "pk_matrix, device against host assembly")
    og = oracle.GSW(ring_o, n, m, sampler=sampler, seed=99, check_max=checked)
    osk, opk = og.keygen()
    _assert_equal(sk.t, osk.t, "t against the oracle")
    _assert_equal(sk.s, osk.s, "s")
    _assert_equal(sk.v, osk.v, "v")
    _assert_equal(pk.pk_matrix, opk.pk_matrix, "pk_matrix against the oracle")
    rbits = np.stack([og.random_bits() for _ in range(2)])
    ct = gsw.encrypt_batch(pk, [1, 0], rbits=rbits)
    for b, mu in enumerate((1, 0)):
        _assert_equal(ct.to_compact()[b], og.encrypt_compact(opk, mu, rbits[b]), f"encryption under the device key [{b}]")
    assert gsw.decrypt_batch(sk, ct).tolist() == [1, 0]
    # without the download the key still encrypts
    sk3, pk3 = tfhe_amd.GSW(ring_d, n, m, sampler=sampler, seed=99, check_max=checked).keygen(download=False)
    assert pk3.pk_matrix is None and np.array_equal(sk3.t, sk.t)


def test_encrypt_matches_oracle(oracle, tfhe_amd):
    """Batched encryption product R * pk + mu * G against the oracle, R given on the host."""
    for ring_name, n in (("fp", 10), ("z32", 7), ("fp", 8)):
        ring_o, ring_d = _rings(oracle, tfhe_amd)[ring_name]
        m = n * ring_d.ell
        og = oracle.GSW(ring_o, n, m, seed=5, check_max=False)
        sk, pk = og.keygen()
        gsw = tfhe_amd.GSW(ring_d, n, m, seed=5, check_max=False)
        dpk = gsw.make_pk(pk.pk_matrix[:, 1:], oracle.matmul(ring_o, pk.pk_matrix, sk.s.reshape(-1, 1)).reshape(-1),
                          sk.t)
        _assert_equal(dpk.pk_matrix, pk.pk_matrix, "pk_matrix = [B t + e | B]")
        msgs = [0, 1, 12345 % ring_o.q]
        rbits = np.stack([og.random_bits() for _ in msgs])
        ct = gsw.encrypt_batch(dpk, msgs, rbits=rbits)
        got = ct.to_compact()
        for b, mu in enumerate(msgs):
            _assert_equal(got[b], og.encrypt_compact(pk, mu, rbits[b]), f"encrypt[{b}] {ring_name}")
        # dense form of the flattened ciphertext vs the oracle's bit_decomp
        _assert_equal(ct.to_dense(1), oracle.bit_decomp_matrix(ring_o, got[1]), "to_dense")


def _stream_words(seed, stream, counters):
    """The generator of the encryption bits (two rounds of the splitmix64 finaliser over seed, stream and draw
    number), in numpy, for whole arrays of draw numbers; spot-checked against the oracle's scalar function."""
    def mix(z):
        z = z + np.uint64(0x9E3779B97F4A7C15)
        z = (z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
        z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
        return z ^ (z >> np.uint64(31))
    with np.errstate(over="ignore"):
        key = mix(np.uint64(seed) ^ (np.uint64(stream) * np.uint64(0xD1342543DE82EF95)))
        return mix(key + counters.astype(np.uint64))


@pytest.mark.parametrize("ring_name,n,extra", [("fp", 4, 0), ("fp", 10, 3), ("z32", 5, 0), ("z8", 9, 5), ("fp", 13, 30)])
def test_device_random_bits_are_the_documented_stream(oracle, tfhe_amd, ring_name, n, extra):
    """Every word of rnd_dmatrix(N, m, 0, 1) as the device draws it: word (b, row, j) = the low ell bits of draw
    (b N + row) words + j, the last word of a row cut to the bits m leaves it, padding words zero -- into an
    aligned buffer (16-byte stores) and into one that is not (word stores)."""
    from tfhe_amd import _lib

    ring_o, ring_d = _rings(oracle, tfhe_amd)[ring_name]
    ell = ring_d.ell
    m = n * ell + extra
    gsw = tfhe_amd.GSW(ring_d, n, m, seed=77)
    eng = gsw.engine
    words = -(-m // ell)
    ld_r = eng.rbits_elems // eng.big_n
    batch = 3
    draws = np.arange(batch * eng.big_n * words, dtype=np.uint64).reshape(batch, eng.big_n, words)
    expect = (_stream_words(77, 3, draws) & np.uint64((1 << ell) - 1)).astype(np.uint32)
    last_bits = m - (words - 1) * ell
    expect[:, :, -1] &= np.uint32((1 << last_bits) - 1)
    for (b, i, j) in [(0, 0, 0), (1, 5, 1), (2, eng.big_n - 1, words - 2)]:
        assert int(expect[b, i, j]) == oracle.rng_u64(77, 3, (b * eng.big_n + i) * words + j) & ((1 << ell) - 1)
    for shift in (0, 1):  # destination 16-byte aligned / 4 bytes past it
        buf = eng.alloc_u32(eng.rbits_elems * batch + 4)
        _lib.check(_lib.lib().gsw_memset_zero(eng.handle, buf.ptr, 4 * (eng.rbits_elems * batch + 4)))
        _lib.check(_lib.lib().gsw_random_rbits(eng.handle, 77, 3, buf.ptr + 4 * shift, batch))
        eng.sync()
        raw = buf.download(np.uint32, eng.rbits_elems * batch + 4)
        got = raw[shift:shift + eng.rbits_elems * batch].reshape(batch, eng.big_n, ld_r)
        _assert_equal(got[:, :, :words], expect, f"random words {ring_name} n={n} m={m} shift={shift}")
        assert not got[:, :, words:].any() and not raw[:shift].any() and not raw[shift + eng.rbits_elems * batch:].any()
    if extra == 0 and ell > 8:
        density = np.unpackbits(expect.view(np.uint8)).mean() * 32 / ell
        assert 0.49 < density < 0.51


@pytest.mark.parametrize("ring_name,n,extra,batch,tight", [("fp", 10, 0, 70, False), ("fp", 10, 3, 9, True),
                                                           ("z32", 7, 0, 45, False), ("z8", 33, 2, 5, False)])
def test_seeded_encryption_equals_drawn_bits_then_encrypt(oracle, tfhe_amd, ring_name, n, extra, batch, tight):
    """gsw_encrypt_batch_seeded (bits drawn chunk by chunk beside the contraction, never stored whole) gives the
    ciphertexts of gsw_random_rbits + gsw_encrypt_batch bit for bit -- and therefore the oracle's, whose encryption
    of given bits the test above pins -- and the prepared forms gsw_prepare_rhs makes of them; they decrypt."""
    from tfhe_amd import _lib

    L = _lib.lib()
    ring_o, ring_d = _rings(oracle, tfhe_amd)[ring_name]
    m = n * ring_d.ell + extra
    gsw = tfhe_amd.GSW(ring_d, n, m, seed=31, check_max=ring_name != "z32")
    eng = gsw.engine
    sk, pk = gsw.keygen()
    rng = np.random.default_rng(batch)
    msgs = rng.integers(0, 2, size=batch).astype(np.uint32)
    d_m = eng.alloc_u32(batch)
    d_m.upload(msgs)
    d_r = eng.alloc_u32(eng.rbits_elems * batch)
    _lib.check(L.gsw_random_rbits(eng.handle, 31, 9, d_r.ptr, batch))
    want = tfhe_amd.Ciphertext(eng, batch)
    _lib.check(L.gsw_encrypt_batch(eng.handle, pk.prepared.ptr, d_r.ptr, eng.rbits_elems, d_m.ptr, want.ptr, want.stride,
                                   batch))
    want_prep = eng.alloc(eng.prepared_bytes * batch)
    _lib.check(L.gsw_prepare_rhs(eng.handle, want.ptr, want.stride, want_prep.ptr, batch))
    if tight:  # room for one bit matrix per workspace half: chunks of one encryption
        limit = max(2 * 4 * eng.rbits_elems, eng.prepared_bytes)
        assert limit < 4 * 4 * eng.rbits_elems
        _lib.check(L.gsw_ctx_set_workspace_limit(eng.handle, limit))
    got = tfhe_amd.Ciphertext(eng, batch)
    got_prep = eng.alloc(eng.prepared_bytes * batch)
    _lib.check(L.gsw_encrypt_batch_seeded(eng.handle, pk.prepared.ptr, 31, 9, d_m.ptr, got.ptr, got.stride, got_prep.ptr,
                                          batch))
    eng.sync()
    _assert_equal(got.to_compact(), want.to_compact(), f"seeded encryption {ring_name} n={n} m={m} batch={batch}")
    assert np.array_equal(got_prep.download(np.uint8, eng.prepared_bytes * batch),
                          want_prep.download(np.uint8, eng.prepared_bytes * batch))
    assert gsw.decrypt_batch(sk, got).tolist() == msgs.tolist()
    # without the prepared output, twice in a row on the same workspace, and a product right behind it on the
    # same streams (its preparation reuses the workspace the bits lived in)
    for _ in range(2):
        again = tfhe_amd.Ciphertext(eng, batch)
        _lib.check(L.gsw_encrypt_batch_seeded(eng.handle, pk.prepared.ptr, 31, 9, d_m.ptr, again.ptr, again.stride, None,
                                              batch))
    prod = gsw.mult(again, got)
    og = oracle.GSW(ring_o, n, m)
    c = want.to_compact()
    _assert_equal(again.to_compact(), c, "second seeded run")
    _assert_equal(prod.to_compact()[batch - 1], og.mul_gadget(c[batch - 1], c[batch - 1]), "product after encryption")
    # through the Python front end: the same stream numbering as explicit bits would get
    assert gsw.decrypt_batch(sk, gsw.encrypt_batch(pk, msgs.tolist())).tolist() == msgs.tolist()


def test_mp_decrypt(oracle, tfhe_amd):
    """gsw_impl.rs:189-207: full-width messages over Zpow2<31>; Fp is refused like the reference."""
    ring_o, ring_d = _rings(oracle, tfhe_amd)["z31"]
    gsw = tfhe_amd.GSW(ring_d, 10, 310, seed=3)
    sk, pk = gsw.keygen()
    msgs = oracle.fill_uniform(ring_o, 123, 0, 0, ring_o.max_u64() - 1, 6)
    ct = gsw.encrypt_batch(pk, msgs)
    assert gsw.mp_decrypt_batch(sk, ct).tolist() == msgs.tolist()
    og = oracle.GSW(ring_o, 10, 310)
    osk = oracle.SecretKey(sk.t, sk.s, sk.v)
    assert og.mp_decrypt_compact(osk, ct.to_compact()[2]) == int(msgs[2])
    fp = tfhe_amd.GSW(tfhe_amd.Fp(), 2, 62)
    fsk, fpk = fp.keygen()
    with pytest.raises(RuntimeError, match="Only supported for pow2 rings"):
        fp.mp_decrypt(fsk, fp.encrypt(fpk, 1))


def test_errors_mirror_the_reference_panics(tfhe_amd):
    with pytest.raises(AssertionError, match="max <= P"):  # error_sampling.rs:21 via gsw_impl.rs:18
        tfhe_amd.GSW(tfhe_amd.Zpow2(32), 4, 128).keygen()
    g = tfhe_amd.GSW(tfhe_amd.Fp(), 2, 62)
    with pytest.raises(ValueError, match="different sizes"):  # gsw_impl.rs:105-106
        g.mult_dense(np.zeros((3, 3)), np.zeros((4, 4)))
    with pytest.raises(ValueError, match="Dimension Mismatch"):  # pk.rs:24-25
        g.make_pk(np.zeros((62, 2)), np.zeros(61), np.zeros(2))
    with pytest.raises(tfhe_amd.GswError):
        tfhe_amd.Engine(tfhe_amd.Zpow2(40), 2, 80)
    for absent in ("12", "6", "0"):  # timing-only and superseded kernels: absent from the shipped library
        os.environ["GSW_GEMM_VARIANT"] = absent
        try:
            with pytest.raises(tfhe_amd.GswError, match="GSW_GEMM_VARIANT"):
                tfhe_amd.Engine(tfhe_amd.Fp(), 2, 62)
        finally:
            os.environ.pop("GSW_GEMM_VARIANT", None)


@pytest.mark.parametrize("ring_name", ["fp", "z30"])
def test_decrypt_after_mul_and_nand(oracle, tfhe_amd, ring_name):
    """Every product also decrypts to mu1 * mu2 (1 - mu1 * mu2 for NAND)."""
    ring_o, ring_d = _rings(oracle, tfhe_amd)[ring_name]
    n = 10
    gsw = tfhe_amd.GSW(ring_d, n, n * ring_d.ell, seed=11)
    og = oracle.GSW(ring_o, n, n * ring_o.ell)
    sk, pk = gsw.keygen()
    mu1 = [0, 0, 1, 1]
    mu2 = [0, 1, 0, 1]
    c1, c2 = gsw.encrypt_batch(pk, mu1), gsw.encrypt_batch(pk, mu2)
    prod, nand = gsw.mult(c1, c2), gsw.nand(c1, c2)
    assert gsw.decrypt_batch(sk, prod).tolist() == [a * b for a, b in zip(mu1, mu2)]
    assert gsw.decrypt_batch(sk, nand).tolist() == [1 - a * b for a, b in zip(mu1, mu2)]
    h1, h2 = c1.to_compact(), c2.to_compact()
    _assert_equal(prod.to_compact(), og.mul_gadget_batch(h1, h2), "product of real ciphertexts")
    # the dense product of the flattened operands is the same ciphertext (gsw_impl.rs:107)
    dense = gsw.mult_dense(c1.to_dense(3), c2.to_dense(3))
    _assert_equal(gsw.bit_decomp_inv_matrix(dense, quirk=False), prod.to_compact()[3], "dense vs compact product")
    # add (gsw_impl.rs:87-94) and mult_const (gsw_impl.rs:97-101) in compact form
    s = gsw.add(c1, c2)
    _assert_equal(s.to_compact(), og.add(h1.reshape(4, -1), h2.reshape(4, -1)).reshape(h1.shape), "add")
    gsw.mult_const(s, 3)
    _assert_equal(s.to_compact(), og.mult_const(og.add(h1.reshape(4, -1), h2.reshape(4, -1)), 3).reshape(h1.shape),
                  "mult_const")


def test_depth8_chain_noise(oracle, tfhe_amd):
    """acc <- mult(acc, fresh_i), depth 8, sigma = 1: the product stays decryptable and the noise of
    row ell-1 stays far below q/4 (additive growth with the accumulator as the decomposed operand)."""
    ring_o, ring_d = _rings(oracle, tfhe_amd)["fp"]
    n = 10
    gsw = tfhe_amd.GSW(ring_d, n, 310, seed=2025)
    og = oracle.GSW(ring_o, n, 310)
    sk, pk = gsw.keygen()
    osk = oracle.SecretKey(sk.t, sk.s, sk.v)
    for msgs in ([1] * 9, [1, 1, 1, 0, 1, 1, 1, 1, 1]):
        fresh = gsw.encrypt_batch(pk, msgs)
        host = fresh.to_compact()
        acc = tfhe_amd.Ciphertext.from_compact(gsw.engine, host[0])
        acc_o, want = host[0], msgs[0]
        for i in range(1, 9):
            acc = gsw.mult(acc, tfhe_amd.Ciphertext.from_compact(gsw.engine, host[i]))
            acc_o = og.mul_gadget(acc_o, host[i])
            want *= msgs[i]
            assert gsw.decrypt(sk, acc) == want
        _assert_equal(acc.to_compact(), acc_o, "depth-8 chain")
        noise = og.noise(osk, acc_o, want)
        assert noise < 20000 < ring_o.q // 4, noise  # ~ 8 * sqrt(N/2) * sqrt(m/2) = 1.3e3 expected


# ---------------------------------------------------------------- dense reference-format operations
@pytest.mark.parametrize("ring_name", ["fp", "z32", "z8"])
def test_dense_ops_arbitrary_entries(oracle, tfhe_amd, ring_name):
    """mult/nand/add/mult_const of gsw_impl.rs:87-123 on arbitrary N x N matrices, and the gadget
    helpers of helper.rs including the `== 1` quirk of bit_decomp_inv."""
    ring_o, ring_d = _rings(oracle, tfhe_amd)[ring_name]
    n = 2
    gsw = tfhe_amd.GSW(ring_d, n, n * ring_d.ell)
    og = oracle.GSW(ring_o, n, n * ring_o.ell)
    dim = ring_o.big_n(n) + 5  # not a multiple of the 16 x 16 tile
    rng = np.random.default_rng(3)
    a = rng.integers(0, ring_o.q, size=(dim, dim), dtype=np.uint64).astype(np.uint32)
    b = rng.integers(0, ring_o.q, size=(dim, dim), dtype=np.uint64).astype(np.uint32)
    _assert_equal(gsw.mult_dense(a, b), og.mult(a, b), "mult_dense")
    assert gsw.last_dense_path() == 0  # small: the plain kernel by default
    _assert_equal(gsw.nand_dense(a, b), og.nand(a, b), "nand_dense")
    # the same on sixteen signed-digit int8 products (MFMA), forced at this size; extreme entries too
    gsw.set_dense_fast(2)
    _assert_equal(gsw.mult_dense(a, b), og.mult(a, b), "mult_dense, digit products")
    assert gsw.last_dense_path() == 2
    _assert_equal(gsw.nand_dense(a, b), og.nand(a, b), "nand_dense, digit products")
    top = np.full((dim, dim), ring_o.q - 1, dtype=np.uint32)
    edge = np.array([v for v in _EDGE_Z32 if v < ring_o.q], dtype=np.uint32)
    e1, e2 = edge[rng.integers(0, len(edge), size=(dim, dim))], edge[rng.integers(0, len(edge), size=(dim, dim))]
    for x, y, what in ((top, top, "all q - 1"), (e1, e2, "edge values"), (top, b, "q - 1 times random"), (a, np.zeros_like(a), "zero")):
        _assert_equal(gsw.mult_dense(x, y), og.mult(x, y), f"digit products, {what}")
        assert gsw.last_dense_path() == 2
    gsw.set_dense_fast(1)
    _assert_equal(gsw.add_dense(a, b), og.add(a, b), "add_dense")
    _assert_equal(gsw.mult_const_dense(a, ring_o.q - 2), og.mult_const(a, ring_o.q - 2), "mult_const_dense")
    comp = rng.integers(0, ring_o.q, size=(7, 5), dtype=np.uint64).astype(np.uint32)
    bits = gsw.bit_decomp_matrix(comp)
    _assert_equal(bits, oracle.bit_decomp_matrix(ring_o, comp), "bit_decomp")
    _assert_equal(gsw.bit_decomp_inv_matrix(bits), comp, "bit_decomp_inv round trip")
    small = rng.integers(0, 4, size=(7, 5 * ring_o.ell), dtype=np.uint64).astype(np.uint32)  # entries 0..3
    _assert_equal(gsw.bit_decomp_inv_matrix(small, quirk=True), oracle.bit_decomp_inv_matrix(ring_o, small, True),
                  "bit_decomp_inv quirk")
    _assert_equal(gsw.bit_decomp_inv_matrix(small, quirk=False), oracle.bit_decomp_inv_matrix(ring_o, small, False),
                  "bit_decomp_inv linear")
    _assert_equal(gsw.flatten_matrix(small), oracle.flatten_matrix(ring_o, small), "flatten")
    ones = np.ones((1, ring_o.ell), dtype=np.uint32)
    assert gsw.bit_decomp_inv_matrix(ones)[0, 0] == oracle.bit_decomp_inv_matrix(ring_o, ones)[0, 0]


@pytest.mark.parametrize("ring_name,dim", [("fp", 1500), ("z32", 1100), ("z30", 1030), ("fp", 64), ("fp", 65), ("z32", 1)])
def test_dense_mult_arbitrary_left_operand_on_digit_products(oracle, tfhe_amd, ring_name, dim):
    """mult / nand (gsw_impl.rs:104-123) with a left operand of arbitrary entries -- what mult(mult(a, b), c) is in
    the reference's default build -- on the digit x digit MFMA products: whole products against the oracle at
    sizes with ragged tiles (dim not a multiple of 64), the largest accumulators the size allows (all entries
    q - 1 in one row band), and the default policy (from 1024 rows up) against the plain kernel."""
    ring_o, ring_d = _rings(oracle, tfhe_amd)[ring_name]
    gsw = tfhe_amd.GSW(ring_d, 2, 2 * ring_d.ell)
    og = oracle.GSW(ring_o, 2, 2 * ring_o.ell)
    rng = np.random.default_rng(dim)
    a = rng.integers(0, ring_o.q, size=(dim, dim), dtype=np.uint64).astype(np.uint32)
    b = rng.integers(0, ring_o.q, size=(dim, dim), dtype=np.uint64).astype(np.uint32)
    a[: min(dim, 20)] = ring_o.q - 1
    b[:, : min(dim, 20)] = ring_o.q - 1
    if dim > 70:
        a[64:70] = 0x7F7F7F7F % ring_o.q  # every digit 127
        b[:, 64:70] = 0x7F7F7F7F % ring_o.q
    expect = oracle.matmul(ring_o, a, b)
    got = gsw.mult_dense(a, b)
    assert gsw.last_dense_path() == (2 if dim >= 1024 else 0)
    _assert_equal(got, expect, f"default policy, dim {dim}")
    gsw.set_dense_fast(2)
    _assert_equal(gsw.mult_dense(a, b), expect, f"digit products, dim {dim}")
    assert gsw.last_dense_path() == 2
    want_nand = ((np.eye(dim, dtype=np.int64) - expect.astype(np.int64)) % ring_o.q).astype(np.uint32)
    _assert_equal(gsw.nand_dense(a, b), want_nand, f"nand, digit products, dim {dim}")
    gsw.set_dense_fast(0)
    _assert_equal(gsw.mult_dense(a, b), expect, "plain kernel")
    assert gsw.last_dense_path() == 0


def test_dense_mult_arbitrary_left_operand_n256_sampled_rows_and_slabs(oracle, tfhe_amd):
    """N = 7967 (n = 256): arbitrary left operand on the digit x digit products, sampled rows against the oracle; the
    same product with a workspace limit that forces the left operand through in several row slabs."""
    from tfhe_amd import _lib

    ring_o, ring_d = _rings(oracle, tfhe_amd)["fp"]
    n = 256
    gsw = tfhe_amd.GSW(ring_d, n, n * 31)
    eng, L = gsw.engine, _lib.lib()
    dim = eng.big_n
    d_a, d_b, d_o, d_o2 = (eng.alloc_u32(dim * dim) for _ in range(4))
    _lib.check(L.gsw_fill_uniform(eng.handle, 21, 1, 0, ring_o.q - 1, 0, d_a.ptr, dim * dim))
    _lib.check(L.gsw_fill_uniform(eng.handle, 21, 2, 0, ring_o.q - 1, 0, d_b.ptr, dim * dim))
    _lib.check(L.gsw_mul_dense(eng.handle, d_a.ptr, d_b.ptr, d_o.ptr, dim))
    assert gsw.last_dense_path() == 2
    a = d_a.download(np.uint32, dim * dim).reshape(dim, dim)
    b = d_b.download(np.uint32, dim * dim).reshape(dim, dim)
    got = d_o.download(np.uint32, dim * dim).reshape(dim, dim)
    rows = [0, 1, 63, 64, 4095, 4096, 4097, dim - 65, dim - 1]
    _assert_equal(got[rows], oracle.matmul(ring_o, a[rows], b), "sampled rows of the N = 7967 product")
    # 256 MB of right-operand planes + room for 1024 left-operand rows at a time: eight slabs
    pad = (dim + 63) // 64 * 64
    _lib.check(L.gsw_ctx_set_workspace_limit(eng.handle, 4 * pad * pad + 4 * pad * 1024))
    _lib.check(L.gsw_nand_dense(eng.handle, d_a.ptr, d_b.ptr, d_o2.ptr, dim))
    assert gsw.last_dense_path() == 2
    got2 = d_o2.download(np.uint32, dim * dim).reshape(dim, dim)
    want = ((np.eye(dim, dtype=np.int64) - got.astype(np.int64)) % ring_o.q).astype(np.uint32)
    _assert_equal(got2, want, "nand through row slabs vs I - product")


@pytest.mark.parametrize("ring_name", ["fp", "z8"])
def test_dense_ops_with_the_use_flatten_feature(oracle, tfhe_amd, ring_name):
    """The reference's non-default cargo feature use_flatten (Cargo.toml:16-17) as a switch of the host mirror: add,
    mult_const, mult and nand then apply flatten_matrix to their result (gsw_impl.rs:91-92,99-100,108-109,120-121),
    with the `== 1` quirk of bit_decomp_inv (helper.rs:31-35)."""
    ring_o, ring_d = _rings(oracle, tfhe_amd)[ring_name]
    n = 2
    gsw = tfhe_amd.GSW(ring_d, n, n * ring_d.ell, use_flatten=True)
    og = oracle.GSW(ring_o, n, n * ring_o.ell)
    dim = ring_o.big_n(n)
    rng = np.random.default_rng(31)
    a = rng.integers(0, 2, size=(dim, dim), dtype=np.uint64).astype(np.uint32)  # small entries: products hit 0, 1, 2, ..
    b = rng.integers(0, 2, size=(dim, dim), dtype=np.uint64).astype(np.uint32)
    a[np.arange(dim), np.arange(dim)] += 1
    flat = lambda x: oracle.flatten_matrix(ring_o, x)  # noqa: E731
    _assert_equal(gsw.mult_dense(a, b), flat(og.mult(a, b)), "mult + flatten")
    _assert_equal(gsw.nand_dense(a, b), flat(og.nand(a, b)), "nand + flatten")
    _assert_equal(gsw.add_dense(a, b), flat(og.add(a, b)), "add + flatten")
    _assert_equal(gsw.mult_const_dense(a, 3 % ring_o.q), flat(og.mult_const(a, 3 % ring_o.q)), "mult_const + flatten")


@pytest.mark.parametrize("ring_name,n", [("fp", 10), ("z32", 7), ("z8", 9), ("fp", 40)])
def test_dense_mult_of_reference_format_ciphertexts_runs_on_mfma(oracle, tfhe_amd, ring_name, n):
    """FheScheme::mult / nand (gsw_impl.rs:104-123) on the reference's own N x N ciphertexts,
    BitDecomp(R * A) + mu * I (gsw_impl.rs:27-42): the left operand is 0 / 1 off the diagonal, so the product
    takes the MFMA contraction (path 1) and must equal the oracle's plain N^3 product; the right operand is
    arbitrary (a previous product included).  Left operands that do not fit the form take the digit x digit MFMA
    products (path 2), or the plain kernel when the MFMA kernels are switched off."""
    ring_o, ring_d = _rings(oracle, tfhe_amd)[ring_name]
    m = n * ring_o.ell
    og = oracle.GSW(ring_o, n, m, seed=n, check_max=False)
    gsw = tfhe_amd.GSW(ring_d, n, m, seed=n, check_max=False)
    sk, pk = og.keygen()
    big_n = ring_o.big_n(n)
    rng = np.random.default_rng(n)
    # default policy: the MFMA path from 1024 rows up; mode 2 takes it whenever the form fits
    c1 = og.encrypt(pk, 1)
    _assert_equal(gsw.mult_dense(c1, c1), og.mult(c1, c1), "default policy")
    assert gsw.last_dense_path() == (1 if big_n >= 1024 else 0)
    gsw.set_dense_fast(2)
    for mu1, mu2 in ((1, 1), (0, 1), (5 % ring_o.q, 1), (ring_o.q - 1, 0)):
        c1, c2 = og.encrypt(pk, mu1), og.encrypt(pk, mu2)
        _assert_equal(gsw.mult_dense(c1, c2), og.mult(c1, c2), f"mult, messages {mu1}, {mu2}")
        assert gsw.last_dense_path() == 1
        _assert_equal(gsw.nand_dense(c1, c2), og.nand(c1, c2), f"nand, messages {mu1}, {mu2}")
        assert gsw.last_dense_path() == 1
    # the right operand may be anything: a product (entries up to q - 1), uniform noise
    c1 = og.encrypt(pk, 1)
    anything = rng.integers(0, ring_o.q, size=(big_n, big_n), dtype=np.uint64).astype(np.uint32)
    _assert_equal(gsw.mult_dense(c1, anything), og.mult(c1, anything), "arbitrary right operand")
    assert gsw.last_dense_path() == 1
    if ring_name == "fp" and n == 10:  # decrypts like the reference's product (gsw_impl.rs:48-52)
        osk = sk
        prod = gsw.mult_dense(og.encrypt(pk, 1), og.encrypt(pk, 1))
        assert og.decrypt(osk, prod) == 1
    # one entry off the diagonal that is not 0 / 1: the digit x digit products, same product
    if ring_o.q > 2:
        odd = c1.copy()
        odd[3, big_n - 1] = 2
        _assert_equal(gsw.mult_dense(odd, anything), og.mult(odd, anything), "misfit left operand")
        assert gsw.last_dense_path() == 2
    # extreme diagonals and an all-ones left operand
    ones = np.ones((big_n, big_n), dtype=np.uint32)
    np.fill_diagonal(ones, ring_o.q - 1)
    _assert_equal(gsw.nand_dense(ones, anything), og.nand(ones, anything), "all ones, diagonal q - 1")
    assert gsw.last_dense_path() == 1
    gsw.set_dense_fast(0)
    _assert_equal(gsw.mult_dense(c1, anything), og.mult(c1, anything), "plain kernel")
    assert gsw.last_dense_path() == 0


@pytest.mark.parametrize("ring_name", ["fp", "z32", "z30", "z8", "z1"])
def test_gadget_decomposition_in_every_output_format(oracle, tfhe_amd, ring_name):
    """G^-1 (helper.rs:7-21) as uint32 (the reference's matrix), as bytes, and as a packed bit string, against the
    oracle's bit_decomp_matrix: ragged shapes (rows shorter and longer than one workgroup run, a padded leading
    dimension, one column), extreme values."""
    from tfhe_amd import _lib

    ring_o, ring_d = _rings(oracle, tfhe_amd)[ring_name]
    gsw = tfhe_amd.GSW(ring_d, 2, 2 * ring_d.ell)
    rng = np.random.default_rng(ring_o.ell)
    for rows, cols, ld in ((7, 5, 5), (3, 1, 4), (5, 11, 16), (2, 300, 304), (1, 1500, 1500), (33, 140, 144)):
        a = rng.integers(0, ring_o.q, size=(rows, cols), dtype=np.uint64).astype(np.uint32)
        a[0, :] = ring_o.q - 1
        a[rows - 1, cols - 1] = ring_o.q - 1
        if rows > 1:
            a[1, :] = 0
        expect = oracle.bit_decomp_matrix(ring_o, a)  # rows x cols*ell, 0 / 1
        _assert_equal(gsw.bit_decomp_as(a, _lib.DECOMP_U32, ld), expect, f"u32 {rows}x{cols}")
        _assert_equal(gsw.bit_decomp_as(a, _lib.DECOMP_U8, ld), expect.astype(np.uint8), f"u8 {rows}x{cols}")
        wide = cols * ring_o.ell
        words = (wide + 31) // 32
        padded = np.zeros((rows, words * 32), dtype=np.uint8)
        padded[:, :wide] = expect
        packed = np.packbits(padded, axis=1, bitorder="little").view(np.uint32).reshape(rows, words)
        _assert_equal(gsw.bit_decomp_as(a, _lib.DECOMP_BITS, ld), packed, f"bits {rows}x{cols}")
    with pytest.raises(tfhe_amd.GswError, match="format"):
        gsw.bit_decomp_as(np.zeros((1, 1)), 7)


def test_samplers_and_digest(oracle, tfhe_amd):
    ring_o, ring_d = _rings(oracle, tfhe_amd)["fp"]
    gsw = tfhe_amd.GSW(ring_d, 2, 62, seed=9)
    eng = gsw.engine
    from tfhe_amd import _lib
    L = _lib.lib()
    count = 200000
    buf = eng.alloc_u32(count)
    _lib.check(L.gsw_fill_uniform(eng.handle, 9, 4, 5, P - 1, 0, buf.ptr, count))
    _assert_equal(buf.download(np.uint32, count), oracle.fill_uniform(ring_o, 9, 4, 5, P - 1, count), "uniform")
    _lib.check(L.gsw_sample_naive(eng.handle, 9, 5, buf.ptr, count))
    _assert_equal(buf.download(np.uint32, count), oracle.sample_naive(ring_o, 9, 5, count), "naive sampler")
    _lib.check(L.gsw_sample_gaussian(eng.handle, 9, 6, 1.0, buf.ptr, count))
    g = buf.download(np.uint32, count).astype(np.int64)
    g = np.where(g > P // 2, g - P, g).astype(np.float64)
    assert abs(g.mean()) < 0.02 and abs(g.var() - 1.0) < 0.03 and np.abs(g).max() <= 14
    _lib.check(L.gsw_sample_gaussian(eng.handle, 9, 7, 3.2, buf.ptr, count))
    g = buf.download(np.uint32, count).astype(np.int64)
    g = np.where(g > P // 2, g - P, g).astype(np.float64)
    assert abs(g.var() - 3.2 ** 2) < 0.4
    with pytest.raises(tfhe_amd.GswError):
        _lib.check(L.gsw_fill_uniform(eng.handle, 9, 4, 0, 2 ** 32 - 1, 0, buf.ptr, 4))
    x = oracle.fill_uniform(ring_o, 1, 1, 0, P - 1, 3 * 1000).reshape(3, 1000)
    buf.upload(x)
    dig = eng.alloc(8 * 3)
    _lib.check(L.gsw_digest_batch(eng.handle, buf.ptr, 1000, 1000, 3, dig.ptr))
    assert dig.download(np.uint64, 3).tolist() == [oracle.digest(x[b]) for b in range(3)]


# ---------------------------------------------------------------- benchmark shapes
def _sampled_rows_check(oracle, og, c1, c2, got, rows, what, flags=0):
    for r in rows:
        expect = og.mul_gadget_rows(c1, c2, int(r), 1, flags)[0]
        _assert_equal(got[r], expect, f"{what} row {r}")


@pytest.mark.parametrize("ring_name,n", [("fp", 256), ("fp", 512), ("z32", 512)])
def test_benchmark_shapes_sampled_rows_and_linearity(oracle, tfhe_amd, ring_name, n):
    """n = 256 / 512 (N = 7967 / 15903 / 16416): sampled output rows against the oracle (each row is
    an independent dot-product family), the first and last row tiles included, plus the
    size-independent property mult(C1, X + Y) = mult(C1, X) + mult(C1, Y) on the whole matrix."""
    ring_o, ring_d = _rings(oracle, tfhe_amd)[ring_name]
    gsw = tfhe_amd.GSW(ring_d, n, n * ring_d.ell)
    og = oracle.GSW(ring_o, n, n * ring_o.ell)
    big_n = ring_o.big_n(n)
    rng = np.random.default_rng(n)
    c1, x, y = (_rand_compact(rng, ring_o, n) for _ in range(3))
    d1 = tfhe_amd.Ciphertext.from_compact(gsw.engine, c1)
    dx = tfhe_amd.Ciphertext.from_compact(gsw.engine, x)
    dy = tfhe_amd.Ciphertext.from_compact(gsw.engine, y)
    px = gsw.mult(d1, dx)
    got = px.to_compact()
    rows = np.unique(np.concatenate([[0, 1, 31, 32, 255, 256, big_n - 1, big_n - 2, big_n - 32],
                                     rng.integers(0, big_n, size=24)]))
    _sampled_rows_check(oracle, og, c1, x, got, rows, f"{ring_name} n={n}")
    lhs = gsw.mult(d1, gsw.add(dx, dy)).to_compact()
    rhs = gsw.add(px, gsw.mult(d1, dy)).to_compact()
    _assert_equal(lhs, rhs, "linearity in the right operand")
    nand = gsw.nand(d1, dx).to_compact()
    _sampled_rows_check(oracle, og, c1, x, nand, rows[:6], f"nand {ring_name} n={n}", oracle.FLAG_NAND)


def test_n256_full_product_and_decrypt_after_mul(oracle, tfhe_amd):
    """One full n = 256 product against the oracle (every entry), and decrypt-after-mul on a small
    batch of real encryptions under a real key at that size."""
    ring_o, ring_d = _rings(oracle, tfhe_amd)["fp"]
    n = 256
    gsw = tfhe_amd.GSW(ring_d, n, n * 31, seed=256)
    og = oracle.GSW(ring_o, n, n * 31)
    sk, pk = gsw.keygen()
    mu1, mu2 = [1, 1, 0, 1], [1, 0, 1, 1]
    c1, c2 = gsw.encrypt_batch(pk, mu1), gsw.encrypt_batch(pk, mu2)
    assert gsw.decrypt_batch(sk, c1).tolist() == mu1
    prod = gsw.mult(c1, c2)
    assert gsw.decrypt_batch(sk, prod).tolist() == [a * b for a, b in zip(mu1, mu2)]
    assert gsw.decrypt_batch(sk, gsw.nand(c1, c2)).tolist() == [1 - a * b for a, b in zip(mu1, mu2)]
    h1, h2, hp = c1.to_compact(), c2.to_compact(), prod.to_compact()
    _assert_equal(hp[0], og.mul_gadget(h1[0], h2[0]), "full n=256 product")
    osk = oracle.SecretKey(sk.t, sk.s, sk.v)
    assert og.noise(osk, hp[0], 1) < ring_o.q // 4 // 1000


@pytest.mark.parametrize("ring_name", ["fp", "z32"])
def test_n512_real_key_decrypt_after_mul(oracle, tfhe_amd, ring_name):
    """n = 512 (N = 15903 / 16416): real key, real encryptions, decrypt-after-mul and -nand on 8 products,
    sampled rows of one product against the oracle.  q = 2^32 needs check_max=False: the reference's
    keygen asserts max <= P there (error_sampling.rs:21)."""
    ring_o, ring_d = _rings(oracle, tfhe_amd)[ring_name]
    n = 512
    gsw = tfhe_amd.GSW(ring_d, n, n * ring_d.ell, seed=512, check_max=False)
    og = oracle.GSW(ring_o, n, n * ring_o.ell)
    sk, pk = gsw.keygen()
    mu1 = [1, 1, 0, 0, 1, 1, 0, 1]
    mu2 = [1, 0, 1, 0, 1, 1, 1, 0]
    c1, c2 = gsw.encrypt_batch(pk, mu1), gsw.encrypt_batch(pk, mu2)
    assert gsw.decrypt_batch(sk, c1).tolist() == mu1
    prod, nand = gsw.mult(c1, c2), gsw.nand(c1, c2)
    assert gsw.decrypt_batch(sk, prod).tolist() == [a * b for a, b in zip(mu1, mu2)]
    assert gsw.decrypt_batch(sk, nand).tolist() == [1 - a * b for a, b in zip(mu1, mu2)]
    h1 = tfhe_amd.Ciphertext(gsw.engine, 1, None)
    rows = [0, 31, 255, 256, 8191, ring_o.big_n(n) - 1]
    a = c1.to_compact()[3]
    b = c2.to_compact()[3]
    got = prod.to_compact()[3]
    del h1
    _sampled_rows_check(oracle, og, a, b, got, rows, f"real ciphertexts {ring_name} n=512")
    osk = oracle.SecretKey(sk.t, sk.s, sk.v)
    noise = og.noise(osk, prod.to_compact()[0], 1)
    print(f"\n[{ring_name} n=512] noise after one product: {noise} (q/4 = {ring_o.q // 4})")
    assert noise < ring_o.q // 4 // 1000


def test_depth8_chain_n256(oracle, tfhe_amd):
    """Depth-8 chain acc <- mult(acc, fresh_i) at n = 256 with a real key: every level decrypts, sampled
    rows of every level match the oracle run on the same inputs, and the measured noise of row ell-1 is
    logged against an additive estimate and stays far below q/4.

    Estimate per level: G^-1(acc) has about N/2 ones and multiplies the fresh noise R_i * e, whose
    entries share the mean (sum of e) / 2 because e is one fixed vector of the public key; hence
    |bias| ~ (N/2) * |sum e| / 2 plus a fluctuation of about sqrt(N/2) * sqrt(m/4)."""
    ring_o, ring_d = _rings(oracle, tfhe_amd)["fp"]
    n = 256
    m = n * 31
    gsw = tfhe_amd.GSW(ring_d, n, m, seed=4096)
    og = oracle.GSW(ring_o, n, m)
    sk, pk = gsw.keygen()
    osk = oracle.SecretKey(sk.t, sk.s, sk.v)
    big_n = ring_o.big_n(n)
    e = oracle.matmul(ring_o, pk.pk_matrix, sk.s.reshape(-1, 1)).reshape(-1).astype(np.int64)
    e = np.where(e > ring_o.q // 2, e - ring_o.q, e)
    per_level = (big_n / 2) * abs(int(e.sum())) / 2 + 6 * (big_n / 2 * m / 4) ** 0.5
    for msgs in ([1] * 9, [1, 1, 1, 1, 0, 1, 1, 1, 1]):
        fresh = gsw.encrypt_batch(pk, msgs)
        host = fresh.to_compact()
        acc = tfhe_amd.Ciphertext.from_compact(gsw.engine, host[0])
        want = msgs[0]
        prev = host[0]
        for i in range(1, 9):
            acc = gsw.mult(acc, tfhe_amd.Ciphertext.from_compact(gsw.engine, host[i]))
            want *= msgs[i]
            assert gsw.decrypt(sk, acc) == want
            cur = acc.to_compact()
            _sampled_rows_check(oracle, og, prev, host[i], cur, [30, 4000, big_n - 1], f"chain level {i}")
            noise = og.noise(osk, cur, want)
            print(f"\n[chain n=256 msgs={msgs}] level {i}: noise {noise}, additive estimate {i * per_level:.0f}, "
                  f"q/4 = {ring_o.q // 4}")
            assert noise < 1.5 * i * per_level + 1000 < ring_o.q // 4
            prev = cur


# ---------------------------------------------------------------- signed digits and accumulator start values
_EDGE_FP = [0, 1, 127, 128, 255, 256, 0x7F7F7F7F, 0x7F7F7F80, 0x7F7F7F81, 0x7F800000, 0x7FFFFF7F, 0x7FFFFF80, P - 2, P - 1]
_EDGE_Z32 = _EDGE_FP + [0x80000000, 0x80808080, 0x7FFFFFFF, 0xFFFFFF7F, 0xFFFFFF80, 0xFFFFFFFF]


@pytest.mark.parametrize("wide", [False, True])
@pytest.mark.parametrize("ring_name,n", [("fp", 10), ("fp", 40), ("fp", 71), ("z32", 10), ("z32", 40), ("z31", 33)])
def test_mul_gadget_edge_valued_right_operands(oracle, tfhe_amd, ring_name, n, wide):
    """Right operands made of the values where the signed-digit split changes branch (carries into the top digit,
    the elements of Fp taken as x - p, the words of Z/2^32 that wrap), against left operands of all ones, all
    zeros and random words, so that the limb sums reach both ends of their range; with the 32-bit recombination
    and, forced, with the 64-bit one (the context reports which one its last launch took); mult, nand and the
    fused digest."""
    from tfhe_amd import _lib

    ring_o, ring_d = _rings(oracle, tfhe_amd)[ring_name]
    gsw = tfhe_amd.GSW(ring_d, n, n * ring_d.ell, seed=1)
    og = oracle.GSW(ring_o, n, n * ring_o.ell)
    eng = gsw.engine
    rng = np.random.default_rng(300 + n)
    edges = np.array([v for v in (_EDGE_FP if ring_name == "fp" else _EDGE_Z32) if v < ring_o.q], dtype=np.uint32)
    c1, c2 = _rand_compact(rng, ring_o, n, 4), _rand_compact(rng, ring_o, n, 4)
    c2[0] = edges[rng.integers(0, len(edges), size=c2[0].shape)]
    c2[1] = edges[rng.integers(len(edges) - 4, len(edges), size=c2[1].shape)]  # the largest values only
    c2[2] = ring_o.q - 1
    c1[1] = ring_o.q - 1 if ring_name != "fp" else 0x7FFFFFFE  # p - 1: every bit but the lowest
    c1[2] = 0
    c1[3, ::2] = ring_o.q - 1
    assert gsw.last_combine() == -1
    if wide:
        os.environ["GSW_FORCE_WIDE_COMBINE"] = "1"
    try:
        d1 = tfhe_amd.Ciphertext.from_compact(eng, c1)
        d2 = tfhe_amd.Ciphertext.from_compact(eng, c2)
        got = gsw.mult(d1, d2).to_compact()
        assert gsw.last_combine() == (0 if wide else 1)
        got_nand = gsw.nand(d1, d2).to_compact()
        assert gsw.last_combine() == (0 if wide else 1)
        out, dig = tfhe_amd.Ciphertext(eng, 4), eng.alloc(8 * 4)
        _lib.check(_lib.lib().gsw_mul_gadget_batch_digest(eng.handle, d1.ptr, d1.stride, d2.ptr, d2.stride, out.ptr,
                                                          out.stride, 4, _lib.FLAG_NAND, dig.ptr))
        assert gsw.last_combine() == (0 if wide else 1)
        raw = out.buf.download(np.uint32, eng.compact_elems * 4).reshape(4, -1)
    finally:
        os.environ.pop("GSW_FORCE_WIDE_COMBINE", None)
    assert dig.download(np.uint64, 4).tolist() == [oracle.digest(raw[b]) for b in range(4)]
    _assert_equal(out.to_compact(), got_nand, "nand with the fused digest vs nand")
    for b in range(4):
        _assert_equal(got[b], og.mul_gadget(c1[b], c2[b]), f"mult[{b}] {ring_name} n={n} wide={wide}")
        _assert_equal(got_nand[b], og.mul_gadget(c1[b], c2[b], flags=oracle.FLAG_NAND), f"nand[{b}] {ring_name} n={n}")


@pytest.mark.parametrize("ring_name", ["fp", "z32", "z30"])
def test_n256_rows_with_the_64_bit_recombination(oracle, tfhe_amd, ring_name):
    """The fallback for contractions too long for the 32-bit recombination (more than about 32 000 K slots), forced
    at n = 256 where interior tiles exist (they take the general epilogue then): the context reports the 64-bit form
    for the forced launches and the 32-bit one otherwise; mult, nand and the fused digest of whole products are
    compared between the two forms, sampled rows with the oracle, the digests with the oracle's."""
    from tfhe_amd import _lib

    ring_o, ring_d = _rings(oracle, tfhe_amd)[ring_name]
    n = 256
    gsw = tfhe_amd.GSW(ring_d, n, n * ring_d.ell, seed=1)
    og = oracle.GSW(ring_o, n, n * ring_o.ell)
    eng, L = gsw.engine, _lib.lib()
    rng = np.random.default_rng(77)
    c1, c2 = _rand_compact(rng, ring_o, n), _rand_compact(rng, ring_o, n)
    c1[7] = ring_o.q - 1  # a row of (almost) all ones: the largest limb sums
    d1, d2 = tfhe_amd.Ciphertext.from_compact(eng, c1), tfhe_amd.Ciphertext.from_compact(eng, c2)

    def products():
        mult = gsw.mult(d1, d2).to_compact()
        first = gsw.last_combine()
        out, dig = tfhe_amd.Ciphertext(eng, 1), eng.alloc(8)
        _lib.check(L.gsw_mul_gadget_batch_digest(eng.handle, d1.ptr, d1.stride, d2.ptr, d2.stride, out.ptr, out.stride,
                                                 1, _lib.FLAG_NAND, dig.ptr))
        assert gsw.last_combine() == first
        raw = out.buf.download(np.uint32, eng.compact_elems)
        assert int(dig.download(np.uint64, 1)[0]) == oracle.digest(raw)
        return first, mult, out.to_compact()

    os.environ["GSW_FORCE_WIDE_COMBINE"] = "1"
    try:
        form, wide, wide_nand = products()
    finally:
        os.environ.pop("GSW_FORCE_WIDE_COMBINE", None)
    assert form == 0, "GSW_FORCE_WIDE_COMBINE=1 did not reach the kernel"
    form, lean, lean_nand = products()
    assert form == 1
    _assert_equal(wide, lean, "64-bit vs 32-bit recombination, whole product")
    _assert_equal(wide_nand, lean_nand, "64-bit vs 32-bit recombination, whole nand")
    big_n = ring_o.big_n(n)
    rows = np.concatenate([np.arange(0, 4), [7], rng.integers(0, big_n, 24), np.arange(big_n - 4, big_n)])
    for r in rows:
        _assert_equal(wide[r:r + 1], og.mul_gadget_rows(c1, c2, int(r), 1), f"row {r} vs oracle")
        _assert_equal(wide_nand[r:r + 1], og.mul_gadget_rows(c1, c2, int(r), 1, flags=oracle.FLAG_NAND), f"nand row {r} vs oracle")


def test_n1024_takes_the_64_bit_recombination_natively(oracle, tfhe_amd):
    """n = 1024: 32 896 K slots, beyond the bound of the 32-bit recombination, so the contraction takes start values
    below 2^28 and the 64-bit form by itself; the shape also has a one-column (narrow) tail tile, a thin last row tile
    and a half last stage.  Sampled rows of one product and of one nand against the oracle; the fused digest against
    the oracle's digest of the stored result."""
    import ctypes as C
    from tfhe_amd import _lib

    ring_o, ring_d = _rings(oracle, tfhe_amd)["fp"]
    n = 1024
    out4, lean = (C.c_uint32 * 4)(), C.c_int(-1)
    _lib.check(_lib.lib().gsw_diag_limb_start_values(_lib.RING_FP_M31, 31, 32 * ((n + 1 + 3) // 4 * 4), 0, out4, C.byref(lean)))
    assert lean.value == 0
    gsw = tfhe_amd.GSW(ring_d, n, n * 31, seed=1)
    og = oracle.GSW(ring_o, n, n * 31)
    eng = gsw.engine
    rng = np.random.default_rng(1024)
    c1, c2 = _rand_compact(rng, ring_o, n), _rand_compact(rng, ring_o, n)
    c1[5] = P - 1  # a row of (almost) all ones: the largest limb sums
    d1, d2 = tfhe_amd.Ciphertext.from_compact(eng, c1), tfhe_amd.Ciphertext.from_compact(eng, c2)
    got = gsw.mult(d1, d2).to_compact()
    assert gsw.last_combine() == 0
    out, dig = tfhe_amd.Ciphertext(eng, 1), eng.alloc(8)
    _lib.check(_lib.lib().gsw_mul_gadget_batch_digest(eng.handle, d1.ptr, d1.stride, d2.ptr, d2.stride, out.ptr,
                                                      out.stride, 1, _lib.FLAG_NAND, dig.ptr))
    assert gsw.last_combine() == 0
    assert int(dig.download(np.uint64, 1)[0]) == oracle.digest(out.buf.download(np.uint32, eng.compact_elems))
    got_nand = out.to_compact()
    big_n = ring_o.big_n(n)
    rows = np.concatenate([np.arange(0, 8), rng.integers(0, big_n, 16), np.arange(big_n - 40, big_n, 5)])
    for r in rows:
        _assert_equal(got[r:r + 1], og.mul_gadget_rows(c1, c2, int(r), 1), f"row {r} vs oracle")
        _assert_equal(got_nand[r:r + 1], og.mul_gadget_rows(c1, c2, int(r), 1, flags=oracle.FLAG_NAND), f"nand row {r} vs oracle")


@pytest.mark.parametrize("n,mode", [(271, 2), (272, 1), (679, 1), (680, 0)])
def test_extreme_operands_at_the_ends_of_the_scaling_modes(oracle, tfhe_amd, n, mode):
    """The largest contractions each way of weighting the left operand's bits still takes with the 32-bit
    recombination (bits weighing 2^(k & 3) up to 272 words of K, 2^(k & 1) up to 680, plain beyond), with the operands
    that drive the limb sums to both ends of their range: left operands of all ones, right operands whose digits are
    all -128 or all +127 before and after the scaling.  Sampled rows, mult and nand, against the oracle."""
    from tfhe_amd import _lib

    ring_o, ring_d = _rings(oracle, tfhe_amd)["fp"]
    gsw = tfhe_amd.GSW(ring_d, n, n * ring_d.ell, seed=1)
    og = oracle.GSW(ring_o, n, n * ring_o.ell)
    big_n = ring_o.big_n(n)
    assert _lib.lib().gsw_diag_bit_scale_mode(0, 31, 32 * ((n + 1 + 3) // 4 * 4)) == mode  # K slots: 32 per word, words in fours
    rng = np.random.default_rng(n)
    c1, c2 = _rand_compact(rng, ring_o, n, 4), _rand_compact(rng, ring_o, n, 4)
    c1[0] = 0x7FFFFFFE  # p - 1: every bit but the lowest
    c1[1] = 0x7FFFFFFE
    c1[2, ::2] = 0x7FFFFFFE
    # digits of 0x7F7F7F7F are +127 each, of 0x7F7F7F80 (taken as x - p) the most negative; a value scaled by 2^-c
    # is its rotation, so rotations of those two give extreme digits in the scaled rows
    extremes = np.array([0x7F7F7F7F, 0x7F7F7F80, 0x00808080, 0x7EFEFEFE], dtype=np.uint64)
    rots = []
    for v in extremes:
        for c in range(4):
            r = int(((int(v) << c) | (int(v) >> (31 - c))) & 0x7FFFFFFF)
            rots.append(r if r < ring_o.q else r - ring_o.q)
    rots = np.array(rots, dtype=np.uint32)
    c2[0] = 0x7F7F7F7F
    c2[1] = rots[rng.integers(0, len(rots), size=c2[1].shape)]
    c2[2] = 0x7F7F7F80
    d1 = tfhe_amd.Ciphertext.from_compact(gsw.engine, c1)
    d2 = tfhe_amd.Ciphertext.from_compact(gsw.engine, c2)
    got = gsw.mult(d1, d2).to_compact()
    assert gsw.last_combine() == 1  # the 32-bit recombination
    got_nand = gsw.nand(d1, d2).to_compact()
    rows = np.unique(np.concatenate([[0, 1, 30, 31, 255, 256, big_n - 1, big_n - 31, big_n - 32], rng.integers(0, big_n, size=8)]))
    for b in range(4):
        _sampled_rows_check(oracle, og, c1[b], c2[b], got[b], rows, f"mult[{b}] n={n}")
        _sampled_rows_check(oracle, og, c1[b], c2[b], got_nand[b], rows[:5], f"nand[{b}] n={n}", oracle.FLAG_NAND)
