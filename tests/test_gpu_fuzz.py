"""Seeded differential runs of the C ABI against the CPU oracle on shapes nobody picked by hand: random ring
(F_p or Z/2^M for any M in 1..32), random n, random batch, entries laid out with random gaps between them, mult or
nand, products through the batch call, the prepared-operand call and a gate list.  Bit exact, every entry compared.
The seeds are fixed, so a failure names a case that fails again."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

CASES = int(os.environ.get("GSW_FUZZ_CASES", "48"))  # a soak run sets more; the suite runs the first 48


@pytest.fixture(scope="module")
def pkg():
    import tfhe_amd as pkg
    from tfhe_amd import _lib

    _lib.lib()  # raises if libgsw_hip.so is missing: there is no fallback
    return pkg


def _case(seed):
    rng = np.random.default_rng(0xF022 + seed)
    if rng.integers(0, 3) == 0:
        ring = ("fp", 31)
    else:
        ring = ("z", int(rng.integers(1, 33)))
    # most cases small; a few reach several column tiles and more than one row tile
    n = int(rng.integers(1, 24)) if seed % 4 else int(rng.integers(24, 130))
    batch = int(rng.integers(1, 6))
    return rng, ring, n, batch


def _rings(oracle, pkg, ring):
    if ring[0] == "fp":
        return oracle.fp(), pkg.Fp()
    return oracle.zpow2(ring[1]), pkg.Zpow2(ring[1])


def _upload_spread(eng, host, gap_elems, rng):
    """`host` (batch, N, n + 1) into one device buffer whose entries lie `compact_elems + gap_elems` words apart,
    the gaps filled with noise that no kernel may read as data or overwrite."""
    batch = host.shape[0]
    stride = eng.compact_elems + gap_elems
    buf = eng.alloc_u32(stride * batch)
    image = rng.integers(0, 2**32, size=(batch, stride), dtype=np.uint64).astype(np.uint32)
    for b in range(batch):
        padded = np.zeros((eng.big_n, eng.ld), dtype=np.uint32)
        padded[:, :host.shape[2]] = host[b]
        image[b, :eng.compact_elems] = padded.reshape(-1)
    buf.upload(image.reshape(-1))
    return buf, stride, image


def _download_spread(buf, stride, batch, eng, width):
    raw = buf.download(np.uint32, stride * batch).reshape(batch, stride)
    mats = raw[:, :eng.compact_elems].reshape(batch, eng.big_n, eng.ld)
    return raw, mats[:, :, :width], mats[:, :, width:]


@pytest.mark.parametrize("seed", range(CASES))
def test_products_on_random_shapes_and_layouts(oracle, pkg, seed):
    from tfhe_amd import _lib

    L = _lib.lib()
    rng, ring, n, batch = _case(seed)
    ring_o, ring_d = _rings(oracle, pkg, ring)
    gsw = pkg.GSW(ring_d, n, n * ring_d.ell, seed=seed)
    eng = gsw.engine
    og = oracle.GSW(ring_o, n, n * ring_o.ell)
    width = n + 1
    shape = (batch, ring_o.big_n(n), width)
    c1 = rng.integers(0, ring_o.q, size=shape, dtype=np.uint64).astype(np.uint32)
    c2 = rng.integers(0, ring_o.q, size=shape, dtype=np.uint64).astype(np.uint32)
    if seed % 3 == 0:  # extreme rows: all ones / all zeros in the binary operand, the largest limbs in the other
        c1[0, ::2] = ring_o.q - 1
        c1[0, 1::2] = 0
        c2[0] = ring_o.q - 1
    flags = int(rng.integers(0, 2)) * _lib.FLAG_NAND
    # entries 16-byte aligned as the ABI asks: gaps in multiples of 4 words
    gap1, gap2, gap_out = (4 * int(rng.integers(0, 9)) for _ in range(3))
    d1, s1, _ = _upload_spread(eng, c1, gap1, rng)
    d2, s2, _ = _upload_spread(eng, c2, gap2, rng)
    dout, sout, before = _upload_spread(eng, np.zeros(shape, dtype=np.uint32), gap_out, rng)
    want = np.stack([og.mul_gadget(c1[b], c2[b], flags=flags) for b in range(batch)])

    _lib.check(L.gsw_mul_gadget_batch(eng.handle, d1.ptr, s1, d2.ptr, s2, dout.ptr, sout, batch, flags))
    eng.sync()
    raw, got, padding = _download_spread(dout, sout, batch, eng, width)
    label = f"seed {seed}: {ring} n={n} batch={batch} flags={flags} gaps={gap1},{gap2},{gap_out}"
    assert np.array_equal(got, want), label
    assert not padding.any(), label + ": padding columns"
    assert np.array_equal(raw[:, eng.compact_elems:], before[:, eng.compact_elems:]), label + ": gap overwritten"

    # the same products with the right operands prepared by a separate call
    prep = eng.alloc(eng.prepared_bytes * batch)
    _lib.check(L.gsw_prepare_rhs(eng.handle, d2.ptr, s2, prep.ptr, batch))
    dout.upload(before.reshape(-1))
    _lib.check(L.gsw_mul_gadget_prepared(eng.handle, d1.ptr, s1, prep.ptr, eng.prepared_bytes, dout.ptr, sout,
                                         batch, flags))
    eng.sync()
    _, got, padding = _download_spread(dout, sout, batch, eng, width)
    assert np.array_equal(got, want), label + " (prepared)"
    assert not padding.any(), label + " (prepared): padding columns"

    # broadcast of entry 0 of the right operand (stride 0)
    _lib.check(L.gsw_mul_gadget_batch(eng.handle, d1.ptr, s1, d2.ptr, 0, dout.ptr, sout, batch, flags))
    eng.sync()
    _, got, _ = _download_spread(dout, sout, batch, eng, width)
    for b in range(batch):
        assert np.array_equal(got[b], og.mul_gadget(c1[b], c2[0], flags=flags)), label + f" (broadcast, entry {b})"


@pytest.mark.parametrize("seed", range(16))
def test_gate_lists_on_random_pools(oracle, pkg, seed):
    """A random gate list (mult, nand, add, mult_const over a pool with random operand reuse) against the same list
    walked on the host with the oracle."""
    rng = np.random.default_rng(0x6A7E + seed)
    ring = ("fp", 31) if seed % 2 else ("z", int(rng.integers(2, 33)))
    ring_o, ring_d = _rings(oracle, pkg, ring)
    n = int(rng.integers(1, 20))
    gsw = pkg.GSW(ring_d, n, n * ring_d.ell, seed=seed)
    eng = gsw.engine
    og = oracle.GSW(ring_o, n, n * ring_o.ell)
    slots = int(rng.integers(4, 10))
    host = rng.integers(0, ring_o.q, size=(slots, ring_o.big_n(n), n + 1), dtype=np.uint64).astype(np.uint32)
    pool = pkg.Ciphertext.from_compact(eng, host)
    gates = []
    model = [host[s].copy() for s in range(slots)]
    for _ in range(int(rng.integers(3, 14))):
        kind = ("mult", "nand", "add", "mult_const")[int(rng.integers(0, 4))]
        a, b = int(rng.integers(0, slots)), int(rng.integers(0, slots))
        if kind == "mult_const":
            k = int(rng.integers(0, ring_o.q))
            gates.append((kind, a, k, a))
            model[a] = og.mult_const(model[a], k)
            continue
        out = int(rng.integers(0, slots))
        if kind in ("mult", "nand"):
            while out in (a, b):  # a product may not overwrite its operand
                out = int(rng.integers(0, slots))
            flags = oracle.FLAG_NAND if kind == "nand" else 0
            model[out] = og.mul_gadget(model[a], model[b], flags=flags)
        else:
            model[out] = og.add(model[a], model[b])
        gates.append((kind, a, b, out))
    gsw.eval_circuit(pool, gates)
    eng.sync()
    got = pool.to_compact()
    for s in range(slots):
        assert np.array_equal(got[s], model[s]), f"seed {seed}: {ring} n={n} slot {s} after {gates}"
    # the same list with a cache of prepared right operands beside the pool: same results, and a slot that several
    # gates multiply by is prepared once unless it was overwritten in between
    pool2 = pkg.Ciphertext.from_compact(eng, host)
    cache = pkg.PreparedCache(eng, slots)
    gsw.eval_circuit(pool2, gates, cache=cache)
    eng.sync()
    got = pool2.to_compact()
    for s in range(slots):
        assert np.array_equal(got[s], model[s]), f"seed {seed} (cached): {ring} n={n} slot {s} after {gates}"


@pytest.mark.parametrize("seed", range(16))
def test_dense_products_on_random_sizes(oracle, pkg, seed):
    """mult / nand (gsw_impl.rs:104-123) on square matrices of random size with arbitrary entries, on the digit x digit
    MFMA products and on the plain kernel, against the oracle's exact product; entries drawn from the whole ring,
    from its extremes, or sparse."""
    rng = np.random.default_rng(0xDE75 + seed)
    ring = ("fp", 31) if seed % 2 == 0 else ("z", int(rng.integers(1, 33)))
    ring_o, ring_d = _rings(oracle, pkg, ring)
    dim = int(rng.integers(1, 200)) if seed % 4 else int(rng.integers(200, 700))
    gsw = pkg.GSW(ring_d, 1, ring_d.ell)
    kind = seed % 3
    if kind == 0:
        a = rng.integers(0, ring_o.q, size=(dim, dim), dtype=np.uint64)
        b = rng.integers(0, ring_o.q, size=(dim, dim), dtype=np.uint64)
    elif kind == 1:  # extremes: 0, 1, q - 1, q - 2 and the values around the digit boundaries
        pick = np.array(sorted({0, 1 % ring_o.q, ring_o.q - 1, max(ring_o.q - 2, 0), 127 % ring_o.q, 128 % ring_o.q,
                                0x7F7F7F7F % ring_o.q, 0x7F7F7F80 % ring_o.q, 0x80808080 % ring_o.q}), dtype=np.uint64)
        a = pick[rng.integers(0, len(pick), size=(dim, dim))]
        b = pick[rng.integers(0, len(pick), size=(dim, dim))]
    else:  # sparse left operand with a few large entries
        a = np.zeros((dim, dim), dtype=np.uint64)
        idx = rng.integers(0, dim, size=(max(dim, 4), 2))
        a[idx[:, 0], idx[:, 1]] = rng.integers(0, ring_o.q, size=len(idx), dtype=np.uint64)
        b = rng.integers(0, ring_o.q, size=(dim, dim), dtype=np.uint64)
    a, b = a.astype(np.uint32), b.astype(np.uint32)
    expect = oracle.matmul(ring_o, a, b)
    want_nand = ((np.eye(dim, dtype=np.int64) - expect.astype(np.int64)) % ring_o.q).astype(np.uint32)
    for mode, path in ((2, 2), (0, 0)):
        gsw.set_dense_fast(mode)
        got = gsw.mult_dense(a, b)
        # a left operand that happens to be 0 / 1 off the diagonal takes the binary x digit contraction only when its
        # size is the context's N (never here: N = 2 ell)
        assert gsw.last_dense_path() == path, (seed, mode)
        assert np.array_equal(got, expect), (seed, ring, dim, mode)
        assert np.array_equal(gsw.nand_dense(a, b), want_nand), (seed, ring, dim, mode, "nand")
