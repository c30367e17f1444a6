"""gsw_mul_gadget_batch_static -- the entry point that allocates nothing (prepared operands staged in the
library's static device array) -- against gsw_mul_gadget_batch on the same inputs, bit for bit, through the
C ABI: two calls in a row on one GPU, the second with a smaller input, so that whatever the first call left
in the shared array must not leak into the second.  Small shapes are also checked against the oracle."""
import ctypes as C

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def tfhe_amd():
    import tfhe_amd as pkg
    from tfhe_amd import _lib

    _lib.lib()  # raises if libgsw_hip.so is missing: there is no fallback
    return pkg


def _both(tfhe_amd, gsw, c1, c2, flags=0, stride2=None):
    """(existing entry point's products, static entry point's products) for one batch of host operands."""
    from tfhe_amd import _lib

    L, eng = _lib.lib(), gsw.engine
    d1 = tfhe_amd.Ciphertext.from_compact(eng, c1)
    d2 = tfhe_amd.Ciphertext.from_compact(eng, c2)
    batch = d1.batch
    s2 = d2.stride if stride2 is None else stride2
    ref, got = tfhe_amd.Ciphertext(eng, batch), tfhe_amd.Ciphertext(eng, batch)
    got.buf.upload(np.full(batch * eng.compact_elems, 0xDEADBEEF, dtype=np.uint32))  # a dirty output buffer
    _lib.check(L.gsw_mul_gadget_batch(eng.handle, d1.ptr, d1.stride, d2.ptr, s2, ref.ptr, ref.stride, batch, flags))
    _lib.check(L.gsw_mul_gadget_batch_static(eng.handle, d1.ptr, d1.stride, d2.ptr, s2, got.ptr, got.stride, batch,
                                             flags))
    eng.sync()
    raw_ref = ref.buf.download(np.uint32, batch * eng.compact_elems)
    raw_got = got.buf.download(np.uint32, batch * eng.compact_elems)
    return raw_ref, raw_got, np.atleast_3d(ref.to_compact()).reshape(batch, eng.big_n, gsw.n + 1)


def _rand(rng, q, batch, big_n, width):
    return rng.integers(0, q, size=(batch, big_n, width), dtype=np.uint64).astype(np.uint32)


@pytest.mark.parametrize("ring_name", ["fp", "z32"])
def test_two_calls_in_a_row_second_smaller(oracle, tfhe_amd, ring_name):
    ring_o, ring_d = {"fp": (oracle.fp(), tfhe_amd.Fp()), "z32": (oracle.zpow2(32), tfhe_amd.Zpow2(32))}[ring_name]
    rng = np.random.default_rng(2024)
    # first call: n = 40 (a full column tile + a packed tail), batch 6; second call: n = 10, batch 2 -- fewer and
    # smaller prepared operands in the same array
    for n, batch, flags in ((40, 6, 0), (10, 2, 0), (10, 3, 1)):
        gsw = tfhe_amd.GSW(ring_d, n, n * ring_d.ell, seed=1)
        og = oracle.GSW(ring_o, n, n * ring_o.ell)
        c1 = _rand(rng, ring_o.q, batch, ring_o.big_n(n), n + 1)
        c2 = _rand(rng, ring_o.q, batch, ring_o.big_n(n), n + 1)
        raw_ref, raw_got, products = _both(tfhe_amd, gsw, c1, c2, flags)
        assert np.array_equal(raw_got, raw_ref), (n, batch, flags)  # padding columns included
        for b in range(batch):
            assert np.array_equal(products[b], og.mul_gadget(c1[b], c2[b], flags=flags)), (n, b)


def test_same_context_second_call_smaller_and_broadcast(oracle, tfhe_amd):
    """One context: batch 5, then batch 2 of other data, then one right operand for a batch of 4 (stride 0)."""
    ring_o, ring_d = oracle.fp(), tfhe_amd.Fp()
    n = 10
    gsw = tfhe_amd.GSW(ring_d, n, 310, seed=3)
    og = oracle.GSW(ring_o, n, 310)
    rng = np.random.default_rng(7)
    for batch in (5, 2):
        c1, c2 = _rand(rng, ring_o.q, batch, 341, 11), _rand(rng, ring_o.q, batch, 341, 11)
        raw_ref, raw_got, products = _both(tfhe_amd, gsw, c1, c2)
        assert np.array_equal(raw_got, raw_ref)
        assert np.array_equal(products, og.mul_gadget_batch(c1, c2))
    c1, c2 = _rand(rng, ring_o.q, 4, 341, 11), _rand(rng, ring_o.q, 1, 341, 11)
    raw_ref, raw_got, products = _both(tfhe_amd, gsw, c1, np.repeat(c2, 4, axis=0), stride2=0)
    assert np.array_equal(raw_got, raw_ref)
    assert np.array_equal(products[3], og.mul_gadget(c1[3], c2[0]))


def test_chunks_through_the_array_at_n256_and_no_allocation(oracle, tfhe_amd):
    """n = 256: one prepared operand is 8.8 MB, so 9 products go through the 64 MiB array in two chunks; then
    2 products.  Device memory in use does not change across the calls; sampled rows of products of both chunks
    are also checked against the oracle."""
    import torch

    from tfhe_amd import _lib

    L = _lib.lib()
    gsw = tfhe_amd.GSW(tfhe_amd.Fp(), 256, 256 * 31, seed=9)
    eng = gsw.engine
    assert 1 < _lib.STATIC_SCRATCH_BYTES // eng.prepared_bytes < 9
    pool = tfhe_amd.Ciphertext(eng, 18)
    _lib.check(L.gsw_fill_uniform(eng.handle, 5, 1, 0, 2147483646, 0, pool.ptr, 18 * eng.compact_elems))
    raw = pool.buf.download(np.uint32, 18 * eng.compact_elems).reshape(18, eng.big_n, eng.ld)
    raw[:, :, 257:] = 0  # the operands' padding columns are zero
    pool.buf.upload(raw)
    ref, got = tfhe_amd.Ciphertext(eng, 9), tfhe_amd.Ciphertext(eng, 9)
    ct = eng.compact_elems
    for batch, first in ((9, 0), (2, 5)):
        lhs, rhs = pool.entry_ptr(first), pool.entry_ptr(9 + first)
        _lib.check(L.gsw_mul_gadget_batch(eng.handle, lhs, ct, rhs, ct, ref.ptr, ct, batch, 0))
        _lib.check(L.gsw_mul_gadget_batch_static(eng.handle, lhs, ct, rhs, ct, got.ptr, ct, batch, 0))  # warm
        got.buf.upload(np.zeros(9 * ct, dtype=np.uint32))
        eng.sync()
        free_before = torch.cuda.mem_get_info(0)[0]
        _lib.check(L.gsw_mul_gadget_batch_static(eng.handle, lhs, ct, rhs, ct, got.ptr, ct, batch, 0))
        eng.sync()
        assert torch.cuda.mem_get_info(0)[0] == free_before
        a = ref.buf.download(np.uint32, batch * ct)
        b = got.buf.download(np.uint32, batch * ct)
        assert np.array_equal(a, b), batch
        og = oracle.GSW(oracle.fp(), 256, 256 * 31)
        prod = b.reshape(batch, eng.big_n, eng.ld)
        for e in (0, batch - 1):  # the first chunk and the last
            c1, c2 = raw[first + e, :, :257], raw[9 + first + e, :, :257]
            for r in (0, 30, 4097, eng.big_n - 1):
                assert np.array_equal(prod[e, r:r + 1, :257], og.mul_gadget_rows(c1, c2, r, 1)), (batch, e, r)


def test_two_contexts_share_the_array_without_racing(tfhe_amd):
    """Two contexts (two streams) on one GPU call the entry point back to back with no synchronisation in between:
    A (two chunks), B (two chunks), A again (one chunk).  The library queues the calls on the shared array, so each
    result equals the allocating entry point's on the same operands."""
    from tfhe_amd import _lib

    L = _lib.lib()
    ga = tfhe_amd.GSW(tfhe_amd.Fp(), 256, 256 * 31, seed=1)
    gb = tfhe_amd.GSW(tfhe_amd.Fp(), 256, 256 * 31, seed=2)
    ea, eb = ga.engine, gb.engine
    ct = ea.compact_elems
    pool = tfhe_amd.Ciphertext(ea, 36)
    _lib.check(L.gsw_fill_uniform(ea.handle, 11, 1, 0, 2147483646, 0, pool.ptr, 36 * ct))
    raw = pool.buf.download(np.uint32, 36 * ct).reshape(36, ea.big_n, ea.ld)
    raw[:, :, 257:] = 0
    pool.buf.upload(raw)
    del raw
    ref_a, ref_b, ref_a2 = (tfhe_amd.Ciphertext(ea, k) for k in (9, 9, 3))
    got_a, got_b, got_a2 = (tfhe_amd.Ciphertext(ea, k) for k in (9, 9, 3))
    calls = ((ea, 0, 9, ref_a, got_a), (eb, 18, 9, ref_b, got_b), (ea, 4, 3, ref_a2, got_a2))
    for eng, first, batch, ref, _ in calls:
        _lib.check(L.gsw_mul_gadget_batch(eng.handle, pool.entry_ptr(first), ct, pool.entry_ptr(first + 9), ct,
                                          ref.ptr, ct, batch, 0))
    ea.sync()
    eb.sync()
    for eng, first, batch, _, got in calls:  # no synchronisation between these three
        _lib.check(L.gsw_mul_gadget_batch_static(eng.handle, pool.entry_ptr(first), ct, pool.entry_ptr(first + 9), ct,
                                                 got.ptr, ct, batch, 0))
    ea.sync()
    eb.sync()
    for _, _, batch, ref, got in calls:
        assert np.array_equal(ref.buf.download(np.uint32, batch * ct), got.buf.download(np.uint32, batch * ct)), batch


def test_sizes_beyond_the_array_are_refused(tfhe_amd):
    from tfhe_amd import _lib

    L = _lib.lib()
    eng = tfhe_amd.Engine(tfhe_amd.Fp(), 768, 768 * 31)  # one prepared operand: 77 MB > 64 MiB
    assert eng.prepared_bytes > _lib.STATIC_SCRATCH_BYTES
    buf = eng.alloc_u32(3 * eng.compact_elems)
    ct = eng.compact_elems
    rc = L.gsw_mul_gadget_batch_static(eng.handle, buf.ptr, ct, buf.ptr + 4 * ct, ct, buf.ptr + 8 * ct, ct, 1, 0)
    assert rc == _lib.E_UNSUPPORTED and b"static scratch" in L.gsw_last_error()
    # the usual argument checks still apply
    small = tfhe_amd.Engine(tfhe_amd.Fp(), 4, 124)
    s = small.alloc_u32(2 * small.compact_elems)
    rc = L.gsw_mul_gadget_batch_static(small.handle, s.ptr, 0, s.ptr, 0, s.ptr, small.compact_elems, 1, 0)
    assert rc == _lib.E_INVALID and b"overlaps" in L.gsw_last_error()
    assert L.gsw_mul_gadget_batch_static(small.handle, s.ptr, 0, s.ptr, 0, s.ptr, small.compact_elems, 0, 0) == _lib.OK
