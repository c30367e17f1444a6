"""Determinism of the contraction under load: the same batch of products, run over and over at the benchmark
shapes, must give the same 64-bit digest for every product every time, and the digests fused into the kernel's
epilogue must equal digests computed from the stored products.

What this is for: the pipelined kernel waits for its LDS-DMA loads by count and hands data from wave to wave
through one barrier per stage.  A mistake in that protocol does not show as a wrong result of a small test; it
shows as one product in ten thousand that differs between two identical runs (it did, once: a variant that read a
just-landed right-operand piece right after the barrier).  Parity with the oracle at these shapes is checked
elsewhere (sampled rows, full products); here only "identical, always" over tens of thousands of products.
"""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("ring_name,n,batch,repeats", [
    ("fp", 256, 1024, 40),   # the headline shape: 40 960 products
    ("fp", 512, 128, 12),
    ("z32", 512, 128, 12),
    ("fp", 10, 4096, 30),    # many small products per launch, operands warm in L2: 122 880 products
])
def test_repeated_batches_give_identical_digests(ring_name, n, batch, repeats):
    import torch

    import tfhe_amd
    from tfhe_amd import _lib

    ring = tfhe_amd.Fp() if ring_name == "fp" else tfhe_amd.Zpow2(32)
    gsw = tfhe_amd.GSW(ring, n, n * ring.ell, seed=99)
    eng, L = gsw.engine, _lib.lib()
    dev = torch.device("cuda", 0)
    ct = eng.compact_elems
    pool = torch.empty((2 * batch, eng.big_n, eng.ld), dtype=torch.int32, device=dev)
    _lib.check(L.gsw_fill_uniform(eng.handle, 99, 0, 0, ring.max_u64(), 1, pool.data_ptr(), 2 * batch * ct))
    pool[:, :, n + 1:] = 0
    out = torch.empty((batch, eng.big_n, eng.ld), dtype=torch.int32, device=dev)
    fused = torch.zeros(batch, dtype=torch.int64, device=dev)
    stored = torch.zeros(batch, dtype=torch.int64, device=dev)
    torch.cuda.synchronize()  # torch fills on its own stream; the engine works on another
    first = {}
    for rep in range(repeats):
        flags = rep & 1  # mult and nand in turn, so that consecutive runs differ in the epilogue
        _lib.check(L.gsw_mul_gadget_batch_digest(eng.handle, pool.data_ptr(), 2 * ct, pool.data_ptr() + 4 * ct, 2 * ct,
                                                 out.data_ptr(), ct, batch, flags, fused.data_ptr()))
        _lib.check(L.gsw_digest_batch(eng.handle, out.data_ptr(), ct, ct, batch, stored.data_ptr()))
        eng.sync()
        d = fused.cpu().numpy().copy()
        assert np.array_equal(d, stored.cpu().numpy()), f"run {rep}: fused digests differ from digests of the stored products"
        if flags not in first:
            first[flags] = d
            continue
        bad = np.nonzero(d != first[flags])[0]
        assert len(bad) == 0, (f"run {rep}: {len(bad)} of {batch} products differ from the first run of the same batch "
                               f"(entries {bad[:8].tolist()})")
    assert len(set(first[0].tolist())) == batch  # digests tell the products apart


@pytest.mark.parametrize("ring_name", ["fp", "z32"])
def test_repeated_dense_products_and_encryptions_are_identical(ring_name):
    """The same check on the other users of the contraction kernel: the reference-format N x N product (flattened left
    operand: slabs of one product as batch entries; arbitrary left operand: digit x digit products) and encryption
    with given random bits (K = m), small shapes whose operands stay in L2, many repeats."""
    import tfhe_amd

    ring = tfhe_amd.Fp() if ring_name == "fp" else tfhe_amd.Zpow2(32)
    n = 10
    gsw = tfhe_amd.GSW(ring, n, n * ring.ell, seed=5, check_max=False)  # Z/2^32: the reference's keygen panics (max <= P)
    big_n = gsw.engine.big_n
    rng = np.random.default_rng(21)
    a = rng.integers(0, ring.q, size=(big_n, big_n), dtype=np.uint64).astype(np.uint32)
    b = rng.integers(0, ring.q, size=(big_n, big_n), dtype=np.uint64).astype(np.uint32)
    flat = gsw.flatten_matrix(a)  # a {0, 1} matrix: takes the binary x digit contraction
    gsw.set_dense_fast(2)  # the MFMA paths at any size
    for left, path, what in ((flat, 1, "flattened left operand"), (a, 2, "arbitrary left operand")):
        first = gsw.mult_dense(left, b)
        assert gsw.last_dense_path() == path, what
        for rep in range(20):
            again = gsw.mult_dense(left, b)
            bad = np.argwhere(again != first)
            assert len(bad) == 0, f"{what}, run {rep}: {len(bad)} entries differ from the first run, first at {bad[0].tolist()}"
    _, pk = gsw.keygen()
    batch = 32
    rbits = rng.integers(0, 2, size=(batch, big_n, n * ring.ell), dtype=np.uint8)
    msgs = [int(v) for v in rng.integers(0, 2, size=batch)]
    first = gsw.encrypt_batch(pk, msgs, rbits=rbits).to_compact()
    for rep in range(20):
        again = gsw.encrypt_batch(pk, msgs, rbits=rbits).to_compact()
        bad = np.argwhere(again != first)
        assert len(bad) == 0, f"encrypt_batch, run {rep}: {len(bad)} entries differ from the first run, first at {bad[0].tolist()}"


def test_seeded_keygen_and_encryption_repeat_exactly():
    """Device key generation and seeded encryption (random bits drawn chunk by chunk on a second stream beside the
    contraction) from the same seed, in fresh contexts: the same key and the same ciphertexts every time."""
    import tfhe_amd

    n, batch = 10, 3000  # several pipeline chunks of random bits
    msgs = [b & 1 for b in range(batch)]
    first = None
    for rep in range(8):
        gsw = tfhe_amd.GSW(tfhe_amd.Fp(), n, n * 31, seed=77)
        sk, pk = gsw.keygen()
        ct = gsw.encrypt_batch(pk, msgs).to_compact()
        now = (np.asarray(sk.t).copy(), ct)
        if first is None:
            first = now
            continue
        assert np.array_equal(now[0], first[0]), f"run {rep}: another secret key from the same seed"
        bad = np.argwhere(now[1] != first[1])
        assert len(bad) == 0, f"run {rep}: {len(bad)} ciphertext entries differ from the first run, first at {bad[0].tolist()}"
