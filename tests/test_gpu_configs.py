"""Every benchmark configuration at its real per-GPU size on one GPU, and the parity checks that need the
large shapes: full n = 512 products (every entry) for Fp and Z/2^32 against the oracle, 2 048 / 4 096
products of n = 512 through slot lists with digests checked over the whole batch, threshold partial
decryptions against the oracle's row dots, batches longer than one grid dimension.

The reference holds no vectors for mult/nand ("parity unpinned"): the oracle's restatement of
gsw_impl.rs:104-123 is the check.
"""
import ctypes as C
import os
from concurrent.futures import ThreadPoolExecutor

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

THREADS = int(os.environ.get("OMP_NUM_THREADS", "0")) or min(16, os.cpu_count() or 1)


@pytest.fixture(scope="module")
def tfhe_amd():
    import tfhe_amd as pkg
    from tfhe_amd import _lib

    _lib.lib()  # raises if libgsw_hip.so is missing: there is no fallback
    return pkg


def _rings(oracle, tfhe_amd, name):
    return {"fp": (oracle.fp(), tfhe_amd.Fp()), "z32": (oracle.zpow2(32), tfhe_amd.Zpow2(32)),
            "z30": (oracle.zpow2(30), tfhe_amd.Zpow2(30))}[name]


def _full_product(og, c1, c2, flags=0):
    """Every row of the oracle's product, row ranges spread over host threads (the C oracle releases the GIL)."""
    rows = c1.shape[0]
    step = (rows + THREADS - 1) // THREADS
    spans = [(r0, min(step, rows - r0)) for r0 in range(0, rows, step)]
    with ThreadPoolExecutor(THREADS) as ex:
        parts = list(ex.map(lambda s: og.mul_gadget_rows(c1, c2, s[0], s[1], flags), spans))
    return np.concatenate(parts, axis=0)


def _assert_equal(got, expect, what):
    if not np.array_equal(got, expect):
        bad = np.argwhere(got != expect)
        raise AssertionError(f"{what}: {len(bad)} of {got.size} entries differ; first at {bad[0].tolist()}: "
                             f"got {got[tuple(bad[0])]}, expect {expect[tuple(bad[0])]}")


@pytest.mark.parametrize("ring_name", ["fp", "z32"])
def test_n512_full_product_every_entry(oracle, tfhe_amd, ring_name):
    """One full n = 512 product (N = 15903 / 16416), every entry, mult and nand.  The operands mix uniform rows
    with rows of q - 1 (every bit set: the largest accumulators 255 * N the shape allows), rows of zeros and of ones."""
    ring_o, ring_d = _rings(oracle, tfhe_amd, ring_name)
    n = 512
    gsw = tfhe_amd.GSW(ring_d, n, n * ring_d.ell)
    og = oracle.GSW(ring_o, n, n * ring_o.ell)
    big_n = ring_o.big_n(n)
    rng = np.random.default_rng(512)
    c1 = rng.integers(0, ring_o.q, size=(big_n, n + 1), dtype=np.uint64).astype(np.uint32)
    c2 = rng.integers(0, ring_o.q, size=(big_n, n + 1), dtype=np.uint64).astype(np.uint32)
    c1[5], c1[256], c1[big_n - 1] = ring_o.q - 1, ring_o.q - 1, ring_o.q - 1
    c1[6], c1[7] = 0, 1
    c2[:, 3], c2[:, n] = ring_o.q - 1, ring_o.q - 1
    c2[:, 4] = 0
    d1 = tfhe_amd.Ciphertext.from_compact(gsw.engine, c1)
    d2 = tfhe_amd.Ciphertext.from_compact(gsw.engine, c2)
    _assert_equal(gsw.mult(d1, d2).to_compact(), _full_product(og, c1, c2), f"full n=512 product {ring_name}")
    _assert_equal(gsw.nand(d1, d2).to_compact(), _full_product(og, c1, c2, oracle.FLAG_NAND),
                  f"full n=512 nand {ring_name}")


@pytest.mark.parametrize("ring_name,batch", [("fp", 2048), ("z32", 4096)])
def test_n512_full_per_gpu_batch_through_slot_lists(oracle, tfhe_amd, ring_name, batch):
    """The per-GPU slice of the 8-GPU configuration (2 048 products, n = 512, Fp) and the whole power-of-two
    configuration (4 096 products, n = 512, q = 2^32) as one gate list over a pool of 16 operands: entry b
    multiplies operands (2b mod 16, 2b + 1 mod 16), so the digests must repeat with period 8 over the whole
    batch, and two full products (one from each end) are compared with the oracle bit for bit."""
    from tfhe_amd import _lib

    L = _lib.lib()
    ring_o, ring_d = _rings(oracle, tfhe_amd, ring_name)
    n, operands = 512, 16
    gsw = tfhe_amd.GSW(ring_d, n, n * ring_d.ell, seed=5)
    eng = gsw.engine
    pool = tfhe_amd.Ciphertext(eng, operands + batch)  # 69 GB / 143 GB of the 288 GB
    _lib.check(L.gsw_memset_zero(eng.handle, pool.ptr, 4 * eng.compact_elems * operands))
    rng = np.random.default_rng(78)
    host = rng.integers(0, ring_o.q, size=(operands, eng.big_n, n + 1), dtype=np.uint64).astype(np.uint32)
    for k in range(operands):
        _lib.check(L.gsw_upload_compact(eng.handle, pool.entry_ptr(k), host[k].ctypes.data))
    gates = [("mult", (2 * b) % operands, (2 * b + 1) % operands, operands + b) for b in range(batch)]
    gsw.eval_circuit(pool, gates)
    digests = eng.alloc(8 * batch)
    _lib.check(L.gsw_digest_batch(eng.handle, pool.entry_ptr(operands), eng.compact_elems, eng.compact_elems, batch,
                                  digests.ptr))
    d = digests.download(np.uint64, batch)
    assert len(set(d[:8].tolist())) == 8
    assert np.array_equal(d, np.tile(d[:8], batch // 8))
    bad = C.c_uint64(1)
    _lib.check(L.gsw_validate_compact(eng.handle, pool.entry_ptr(operands), eng.compact_elems, batch, C.byref(bad)))
    assert bad.value == 0  # canonical values, zero padding, in every product
    og = oracle.GSW(ring_o, n, n * ring_o.ell)
    out = np.empty((eng.big_n, n + 1), dtype=np.uint32)
    for b in (5, batch - 2):
        _lib.check(L.gsw_download_compact(eng.handle, out.ctypes.data, pool.entry_ptr(operands + b)))
        k = b % 8
        _assert_equal(out, _full_product(og, host[2 * k], host[2 * k + 1]), f"product {b} of {batch}")


@pytest.mark.parametrize("ring_name", ["fp", "z30"])
def test_partial_decryptions_match_the_oracle(oracle, tfhe_amd, ring_name):
    """Threshold hooks against the oracle: party i's partial is <row ell-1 of Cc, s_i> as the oracle computes it
    from the same share, and the recombined bit is the oracle's is_zero_one (gsw_impl.rs:127-131) of the
    oracle's sum of partials."""
    ring_o, ring_d = _rings(oracle, tfhe_amd, ring_name)
    n = 10
    gsw = tfhe_amd.GSW(ring_d, n, n * ring_d.ell, seed=32)
    og = oracle.GSW(ring_o, n, n * ring_o.ell)
    sk, pk = gsw.keygen()
    msgs = [0, 1, 1, 0, 1, 0]
    ct = gsw.nand(gsw.encrypt_batch(pk, msgs), gsw.encrypt_batch(pk, [1, 1, 0, 1, 1, 1]))
    host = ct.to_compact()
    for parties in (1, 3, 4):
        shares = gsw.share_secret(sk, parties, seed=10 + parties)
        partials = np.stack([gsw.part_dec(shares[p], ct) for p in range(parties)])
        sums = []
        for b in range(len(msgs)):
            total = 0
            for p in range(parties):
                share_key = oracle.SecretKey(sk.t, shares[p], sk.v)  # row_dot_s only reads .s
                expect = og.row_dot_s(share_key, host[b], ring_o.ell - 1)
                assert int(partials[p, b]) == expect, (parties, p, b)
                total = ring_o.add(total, expect)
            sums.append(total)
        bits = gsw.combine_partials(partials)
        assert bits.tolist() == [ring_o.is_zero_one(x) for x in sums]
        osk = oracle.SecretKey(sk.t, sk.s, sk.v)
        assert bits.tolist() == [og.decrypt_compact(osk, host[b]) for b in range(len(msgs))]
    # a partial that is not a canonical ring value is refused
    bad = partials.copy()
    bad[0, 0] = ring_o.q if ring_o.q < 2 ** 32 else bad[0, 0]
    if ring_o.q < 2 ** 32:
        with pytest.raises(tfhe_amd.GswError, match="canonical"):
            gsw.combine_partials(bad)


def test_decrypt_batches_longer_than_one_grid_dimension(oracle, tfhe_amd):
    """70 000 entries in one decrypt / partial-decrypt / digest call (a grid dimension holds 65 535): raw row dots
    and digests of sampled entries, the last ones included, against the oracle."""
    from tfhe_amd import _lib

    L = _lib.lib()
    ring_o, ring_d = _rings(oracle, tfhe_amd, "fp")
    n, batch = 10, 70000
    gsw = tfhe_amd.GSW(ring_d, n, 310, seed=6)
    og = oracle.GSW(ring_o, n, 310)
    eng = gsw.engine
    sk, _ = gsw.keygen()
    ct = tfhe_amd.Ciphertext(eng, batch)  # 1.5 GB of uniform words: decryption is a row dot whatever the content
    _lib.check(L.gsw_fill_uniform(eng.handle, 99, 1, 0, ring_o.q - 1, 0, ct.ptr, eng.compact_elems * batch))
    bits, raw = gsw.decrypt_batch(sk, ct, raw=True)
    partial = gsw.part_dec(sk.s, ct)
    assert np.array_equal(partial, raw)
    osk = oracle.SecretKey(sk.t, sk.s, sk.v)
    row = np.empty(eng.ld, dtype=np.uint32)
    for b in (0, 1, 65534, 65535, 65536, 69999):
        off = 4 * (eng.compact_elems * b + (ring_o.ell - 1) * eng.ld)
        _lib.check(L.gsw_memcpy_d2h(eng.handle, row.ctypes.data, ct.ptr + off, 4 * eng.ld))
        cc = np.zeros((eng.big_n, n + 1), dtype=np.uint32)
        cc[ring_o.ell - 1] = row[:n + 1]
        expect = og.row_dot_s(osk, cc, ring_o.ell - 1)
        assert int(raw[b]) == expect and int(bits[b]) == ring_o.is_zero_one(expect), b
    # the digests of the same 70 000 entries in one call, against the oracle's digest of sampled entries
    dig = eng.alloc(8 * batch)
    _lib.check(L.gsw_digest_batch(eng.handle, ct.ptr, ct.stride, eng.compact_elems, batch, dig.ptr))
    got = dig.download(np.uint64, batch)
    entry = np.empty(eng.compact_elems, dtype=np.uint32)
    for b in (0, 65534, 65535, 65536, 69999):
        _lib.check(L.gsw_memcpy_d2h(eng.handle, entry.ctypes.data, ct.entry_ptr(b), 4 * eng.compact_elems))
        assert int(got[b]) == oracle.digest(entry), b


def test_validate_compact_counts_what_breaks_the_invariants(tfhe_amd):
    from tfhe_amd import _lib

    L = _lib.lib()
    for ring, bad_value in ((tfhe_amd.Fp(), 2147483647), (tfhe_amd.Zpow2(8), 256)):
        gsw = tfhe_amd.GSW(ring, 5, 5 * ring.ell, seed=1)
        eng = gsw.engine
        rng = np.random.default_rng(4)
        host = rng.integers(0, ring.q, size=(3, eng.big_n, 6), dtype=np.uint64).astype(np.uint32)
        ct = tfhe_amd.Ciphertext.from_compact(eng, host)
        bad = C.c_uint64(7)
        _lib.check(L.gsw_validate_compact(eng.handle, ct.ptr, ct.stride, 3, C.byref(bad)))
        assert bad.value == 0
        raw = ct.buf.download(np.uint32, eng.compact_elems * 3).reshape(3, eng.big_n, eng.ld)
        raw[0, 0, 0] = bad_value      # not canonical
        raw[1, 4, 2] = 0xFFFFFFFF     # not canonical
        raw[2, eng.big_n - 1, eng.ld - 1] = 1  # non-zero padding
        raw[2, 3, 6] = 5              # non-zero padding
        ct.buf.upload(raw)
        _lib.check(L.gsw_validate_compact(eng.handle, ct.ptr, ct.stride, 3, C.byref(bad)))
        assert bad.value == 4
        _lib.check(L.gsw_validate_compact(eng.handle, ct.ptr, ct.stride, 0, C.byref(bad)))
        assert bad.value == 0


def test_rejected_gate_list_leaves_the_pool_untouched(tfhe_amd):
    """A list whose last gate is invalid is refused before anything is launched."""
    gsw = tfhe_amd.GSW(tfhe_amd.Fp(), 3, 93, seed=2)
    rng = np.random.default_rng(5)
    start = rng.integers(0, 2147483647, size=(6, gsw.engine.big_n, 4), dtype=np.uint64).astype(np.uint32)
    for bad_gate, match in ((("mult_const", 0, 2147483647, 1), "canonical"), (("nand", 2, 3, 3), "own operand"),
                            (("mult", 0, 9, 1), "out of range")):
        pool = tfhe_amd.Ciphertext.from_compact(gsw.engine, start)
        gates = [("mult", 0, 1, 2), ("add", 2, 3, 4), ("mult_const", 4, 3, 4), ("nand", 4, 5, 0), bad_gate]
        with pytest.raises(tfhe_amd.GswError, match=match):
            gsw.eval_circuit(pool, gates)
        assert np.array_equal(pool.to_compact(), start)
