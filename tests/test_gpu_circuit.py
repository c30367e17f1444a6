"""Gate lists over a pool of resident ciphertexts (gsw_eval_circuit) against the oracle applying the
same gates one after another (gsw_impl.rs:87-123 in compact form), bit exact, plus decryption of
every output of a NAND circuit built from real encryptions.

The reference holds no vectors for mult/nand/add/mult_const ("parity unpinned"); the oracle's
restatement and decryption are the checks.
"""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def tfhe_amd():
    import tfhe_amd as pkg
    from tfhe_amd import _lib

    _lib.lib()  # raises if libgsw_hip.so is missing: there is no fallback
    return pkg


def _apply_on_oracle(oracle, og, pool, gates):
    """The list's meaning: one gate after another."""
    pool = pool.copy()
    shape = pool[0].shape
    for op, a, b, out in gates:
        if op == "mult":
            pool[out] = og.mul_gadget(pool[a], pool[b])
        elif op == "nand":
            pool[out] = og.mul_gadget(pool[a], pool[b], flags=oracle.FLAG_NAND)
        elif op == "add":
            pool[out] = og.add(pool[a].reshape(1, -1), pool[b].reshape(1, -1)).reshape(shape)
        else:
            pool[out] = og.mult_const(pool[a].reshape(1, -1).copy(), b).reshape(shape)
    return pool


def _xor_gates(bits, levelised):
    """XOR(a_i, b_i) from four NANDs; inputs in slots i and bits+i, result in slot 5*bits+i."""
    per_bit = []
    for i in range(bits):
        a, b, t, u, v, x = i, bits + i, 2 * bits + i, 3 * bits + i, 4 * bits + i, 5 * bits + i
        per_bit.append([("nand", a, b, t), ("nand", a, t, u), ("nand", b, t, v), ("nand", u, v, x)])
    if not levelised:
        return [g for gates in per_bit for g in gates]
    t = [gates[0] for gates in per_bit]
    uv = [g for gates in per_bit for g in gates[1:3]]
    x = [gates[3] for gates in per_bit]
    return t + uv + x


@pytest.mark.parametrize("ring_name", ["fp", "z32"])
def test_xor_circuit_decrypts_and_matches_oracle(oracle, tfhe_amd, ring_name):
    ring_o, ring_d = {"fp": (oracle.fp(), tfhe_amd.Fp()), "z32": (oracle.zpow2(32), tfhe_amd.Zpow2(32))}[ring_name]
    n, bits = 10, 4
    m = n * ring_d.ell
    # the reference's keygen panics over Zpow2<32> (assert max <= P, error_sampling.rs:21): lifted there
    gsw = tfhe_amd.GSW(ring_d, n, m, seed=77, check_max=ring_name != "z32")
    og = oracle.GSW(ring_o, n, m)
    sk, pk = gsw.keygen()
    a_bits, b_bits = [0, 0, 1, 1], [0, 1, 0, 1]
    fresh = gsw.encrypt_batch(pk, a_bits + b_bits).to_compact()
    start = np.zeros((6 * bits,) + fresh.shape[1:], dtype=np.uint32)
    start[:2 * bits] = fresh
    expect = None
    for levelised in (True, False):
        gates = _xor_gates(bits, levelised)
        pool = tfhe_amd.Ciphertext.from_compact(gsw.engine, start)
        gsw.eval_circuit(pool, gates)
        got = pool.to_compact()
        if expect is None:
            expect = _apply_on_oracle(oracle, og, start, gates)
        assert np.array_equal(got, expect), f"levelised={levelised}"
        out = gsw.decrypt_batch(sk, pool)
        assert out[5 * bits:].tolist() == [x ^ y for x, y in zip(a_bits, b_bits)]
        assert out[2 * bits:3 * bits].tolist() == [1 - x * y for x, y in zip(a_bits, b_bits)]


def test_gate_list_hazards_follow_list_order(oracle, tfhe_amd):
    """Random gates with read-after-write, write-after-read and write-after-write between neighbours,
    in-place adds and constant multiples, on uniform random operands."""
    ring_o, ring_d = oracle.fp(), tfhe_amd.Fp()
    n, slots = 3, 12
    gsw = tfhe_amd.GSW(ring_d, n, n * 31, seed=3)
    og = oracle.GSW(ring_o, n, n * 31)
    rng = np.random.default_rng(9)
    start = rng.integers(0, ring_o.q, size=(slots, ring_o.big_n(n), n + 1), dtype=np.uint64).astype(np.uint32)
    gates = []
    for _ in range(60):
        kind = rng.choice(["mult", "nand", "nand", "mult", "add", "mult_const"])
        a, out = int(rng.integers(slots)), int(rng.integers(slots))
        if kind == "mult_const":
            gates.append((kind, a, int(rng.integers(ring_o.q)), out))
            continue
        b = int(rng.integers(slots))
        if kind != "add":
            while out in (a, b):
                out = int(rng.integers(slots))
        gates.append((kind, a, b, out))
    # explicit neighbours: RAW, WAR, WAW, and a repeated operand
    gates += [("mult", 0, 1, 2), ("mult", 2, 3, 4), ("nand", 5, 6, 3), ("nand", 7, 8, 3), ("mult", 9, 9, 10)]
    pool = tfhe_amd.Ciphertext.from_compact(gsw.engine, start)
    gsw.eval_circuit(pool, gates)
    assert np.array_equal(pool.to_compact(), _apply_on_oracle(oracle, og, start, gates))
    # a second list on the same context reuses the staged slot lists
    gsw.eval_circuit(pool, gates[:7])
    twice = _apply_on_oracle(oracle, og, _apply_on_oracle(oracle, og, start, gates), gates[:7])
    assert np.array_equal(pool.to_compact(), twice)


def test_one_level_longer_than_a_pipeline_chunk(oracle, tfhe_amd):
    """300 independent products in one run: the chunked two-stream path with slot lists."""
    ring_o, ring_d = oracle.zpow2(30), tfhe_amd.Zpow2(30)
    n, inputs, count = 2, 10, 300
    gsw = tfhe_amd.GSW(ring_d, n, n * 30, seed=4)
    og = oracle.GSW(ring_o, n, n * 30)
    rng = np.random.default_rng(10)
    start = np.zeros((inputs + count, ring_o.big_n(n), n + 1), dtype=np.uint32)
    start[:inputs] = rng.integers(0, ring_o.q, size=(inputs,) + start.shape[1:], dtype=np.uint64).astype(np.uint32)
    gates = [("mult", int(rng.integers(inputs)), int(rng.integers(inputs)), inputs + (7 * i) % count)
             for i in range(count)]  # 7 and 300 are coprime: a permutation of the output slots
    pool = tfhe_amd.Ciphertext.from_compact(gsw.engine, start)
    gsw.eval_circuit(pool, gates)
    assert np.array_equal(pool.to_compact(), _apply_on_oracle(oracle, og, start, gates))


def test_shared_right_operands_are_prepared_once(oracle, tfhe_amd):
    """Gates of one run that multiply by the same slot share one prepared operand: the levelised XOR circuit's
    second level has two gates per bit with the same right operand; a fan-out of one operand over a run longer
    than a pipeline chunk prepares it once per chunk.  Results against the oracle as ever."""
    ring_o, ring_d = oracle.zpow2(30), tfhe_amd.Zpow2(30)
    n, bits = 2, 6
    gsw = tfhe_amd.GSW(ring_d, n, n * 30, seed=12)
    og = oracle.GSW(ring_o, n, n * 30)
    rng = np.random.default_rng(12)
    start = rng.integers(0, ring_o.q, size=(6 * bits, ring_o.big_n(n), n + 1), dtype=np.uint64).astype(np.uint32)
    gates = _xor_gates(bits, levelised=True)  # levels of 6, 12 and 6 products; level 2 has 6 distinct right operands
    pool = tfhe_amd.Ciphertext.from_compact(gsw.engine, start)
    prepared0, products0 = gsw.engine.counters()
    gsw.eval_circuit(pool, gates)
    prepared1, products1 = gsw.engine.counters()
    assert products1 - products0 == 4 * bits and prepared1 - prepared0 == 3 * bits
    assert np.array_equal(pool.to_compact(), _apply_on_oracle(oracle, og, start, gates))
    # 300 products by one right operand and by two alternating ones, through the chunked pipeline
    inputs, count = 8, 300
    start = np.zeros((inputs + count, ring_o.big_n(n), n + 1), dtype=np.uint32)
    start[:inputs] = rng.integers(0, ring_o.q, size=(inputs,) + start.shape[1:], dtype=np.uint64).astype(np.uint32)
    for rhs_of in (lambda i: 3, lambda i: 3 + (i & 1)):
        gates = [("nand", i % inputs, rhs_of(i), inputs + i) for i in range(count)]
        pool = tfhe_amd.Ciphertext.from_compact(gsw.engine, start)
        prepared0, _ = gsw.engine.counters()
        gsw.eval_circuit(pool, gates)
        prepared1, _ = gsw.engine.counters()
        assert prepared1 - prepared0 <= 2 * 6  # at most two distinct operands in each of at most six chunks
        assert np.array_equal(pool.to_compact(), _apply_on_oracle(oracle, og, start, gates))


def test_cached_gate_lists_match_the_oracle_and_prepare_each_slot_once(oracle, tfhe_amd):
    """gsw_eval_circuit_cached: the random hazard-laden gate list gives the oracle's pool, every slot is prepared at
    most once between two writes to it, and a rejected list leaves the flags alone."""
    ring_o, ring_d = oracle.fp(), tfhe_amd.Fp()
    n, slots = 3, 12
    gsw = tfhe_amd.GSW(ring_d, n, n * 31, seed=3)
    og = oracle.GSW(ring_o, n, n * 31)
    rng = np.random.default_rng(19)
    start = rng.integers(0, ring_o.q, size=(slots, ring_o.big_n(n), n + 1), dtype=np.uint64).astype(np.uint32)
    gates = []
    for _ in range(80):
        kind = rng.choice(["mult", "nand", "nand", "mult", "add", "mult_const"])
        a, out = int(rng.integers(slots)), int(rng.integers(slots))
        if kind == "mult_const":
            gates.append((kind, a, int(rng.integers(ring_o.q)), out))
            continue
        b = int(rng.integers(slots))
        if kind != "add":
            while out in (a, b):
                out = int(rng.integers(slots))
        gates.append((kind, a, b, out))
    pool = tfhe_amd.Ciphertext.from_compact(gsw.engine, start)
    cache = tfhe_amd.PreparedCache(gsw.engine, slots)
    gsw.eval_circuit(pool, gates[:50], cache=cache)
    gsw.eval_circuit(pool, gates[50:], cache=cache)  # the flags carry over from call to call
    assert np.array_equal(pool.to_compact(), _apply_on_oracle(oracle, og, start, gates))
    # the flags tell the truth: every entry marked valid is the prepared form of its slot -- a fan-out over all
    # slots as right operands, half of them through entries the earlier lists left valid, must agree
    fan = [("mult", 0, s, slots + s) for s in range(slots)]
    big = np.concatenate([pool.to_compact(), np.zeros_like(start)])
    pool2 = tfhe_amd.Ciphertext.from_compact(gsw.engine, big)
    cache2 = tfhe_amd.PreparedCache(gsw.engine, 2 * slots)
    cache2.valid[:slots] = cache.valid
    from tfhe_amd import _lib
    _lib.check(_lib.lib().gsw_memcpy_d2h(gsw.engine.handle, (tmp := np.empty(slots * gsw.engine.prepared_bytes, np.uint8)).ctypes.data,
                                         cache.buf.ptr, tmp.size))
    _lib.check(_lib.lib().gsw_memcpy_h2d(gsw.engine.handle, cache2.buf.ptr, tmp.ctypes.data, tmp.size))
    assert 0 < int(cache.valid.sum()) <= slots
    prepared0, _ = gsw.engine.counters()
    gsw.eval_circuit(pool2, fan, cache=cache2)
    prepared1, _ = gsw.engine.counters()
    assert prepared1 - prepared0 == slots - int(cache.valid.sum())  # only the stale ones
    assert np.array_equal(pool2.to_compact(), _apply_on_oracle(oracle, og, big, fan))
    assert cache2.valid[:slots].all() and not cache2.valid[slots:].any()
    gsw.eval_circuit(pool2, fan, cache=cache2)  # everything valid now: no preparation at all
    assert gsw.engine.counters()[0] == prepared1
    with pytest.raises(tfhe_amd.GswError, match="own operand"):
        gsw.eval_circuit(pool2, fan + [("mult", 1, 2, 1)], cache=cache2)
    assert cache2.valid[:slots].all() and not cache2.valid[slots:].any()  # a rejected list changes nothing


def test_chain_over_fresh_encryptions_in_prepared_form_prepares_nothing(oracle, tfhe_amd):
    ring_o, ring_d = oracle.fp(), tfhe_amd.Fp()
    n, chains, depth = 10, 5, 4
    gsw = tfhe_amd.GSW(ring_d, n, 310, seed=44)
    og = oracle.GSW(ring_o, n, 310)
    sk, pk = gsw.keygen()
    pool = tfhe_amd.Ciphertext(gsw.engine, chains * (2 + depth))
    cache = tfhe_amd.PreparedCache(gsw.engine, pool.batch)
    msgs = [[1, 1, 0, 1, 1]] + [[1, 1, 1, 1, 1]] * (depth - 2) + [[1, 0, 1, 1, 1]]
    gsw.encrypt_batch(pk, [1] * chains, out=pool, slot=0)
    for lvl in range(depth):
        gsw.encrypt_batch(pk, msgs[lvl], out=pool, slot=chains * (2 + lvl), cache=cache)
    before = pool.to_compact()
    gates = []
    for lvl in range(depth):
        src, dst = (0, chains) if lvl % 2 == 0 else (chains, 0)
        gates += [("mult", src + c, chains * (2 + lvl) + c, dst + c) for c in range(chains)]
    prepared0, products0 = gsw.engine.counters()
    gsw.eval_circuit(pool, gates, cache=cache)
    prepared1, products1 = gsw.engine.counters()
    assert prepared1 == prepared0 and products1 - products0 == chains * depth
    assert np.array_equal(pool.to_compact(), _apply_on_oracle(oracle, og, before, gates))
    final = 0 if depth % 2 == 0 else chains
    want = [int(all(msgs[lvl][c] for lvl in range(depth))) for c in range(chains)]
    assert gsw.decrypt_batch(sk, pool, slot=final, count=chains).tolist() == want


def test_writes_into_the_pool_invalidate_its_prepared_cache(oracle, tfhe_amd):
    """A slot overwritten behind a cached gate list -- by encrypt_batch without cache=, by mult_const, by an uncached
    gate list -- loses its flag, so the next cached list prepares it again instead of multiplying by the stale
    prepared operand; a cache cannot serve two pools."""
    ring_o, ring_d = oracle.fp(), tfhe_amd.Fp()
    n = 10
    gsw = tfhe_amd.GSW(ring_d, n, 310, seed=77)
    og = oracle.GSW(ring_o, n, 310)
    sk, pk = gsw.keygen()
    pool = gsw.encrypt_batch(pk, [1, 1, 0, 0, 0, 0])
    cache = tfhe_amd.PreparedCache(gsw.engine, pool.batch, pool=pool)
    gsw.eval_circuit(pool, [("mult", 0, 1, 2)], cache=cache)  # prepares slot 1
    assert cache.valid[1] == 1 and gsw.decrypt_batch(sk, pool, slot=2, count=1)[0] == 1
    gsw.encrypt_batch(pk, [0], out=pool, slot=1)  # slot 1 now encrypts 0, written without the cache
    assert cache.valid[1] == 0
    gsw.eval_circuit(pool, [("mult", 0, 1, 3)], cache=cache)
    host = pool.to_compact()
    assert np.array_equal(host[3], og.mul_gadget(host[0], host[1]))
    assert gsw.decrypt_batch(sk, pool, slot=3, count=1)[0] == 0
    assert cache.valid[1] == 1
    gsw.eval_circuit(pool, [("add", 0, 0, 1)])  # an uncached list writes slot 1
    assert not cache.valid.any()
    gsw.eval_circuit(pool, [("mult", 0, 1, 4)], cache=cache)
    host = pool.to_compact()
    assert np.array_equal(host[4], og.mul_gadget(host[0], host[1]))
    gsw.mult_const(pool, 3)  # in place on the whole batch
    assert not cache.valid.any()
    other = gsw.encrypt_batch(pk, [1] * 6)
    with pytest.raises(ValueError, match="another pool"):
        gsw.eval_circuit(other, [("mult", 0, 1, 2)], cache=cache)
    with pytest.raises(ValueError, match="already has"):
        tfhe_amd.PreparedCache(gsw.engine, pool.batch, pool=pool)


def test_bad_gates_are_rejected(tfhe_amd):
    gsw = tfhe_amd.GSW(tfhe_amd.Fp(), 2, 62, seed=1)
    pool = tfhe_amd.Ciphertext(gsw.engine, 4)
    with pytest.raises(tfhe_amd.GswError, match="own operand"):
        gsw.eval_circuit(pool, [("mult", 0, 1, 0)])
    with pytest.raises(tfhe_amd.GswError, match="out of range"):
        gsw.eval_circuit(pool, [("nand", 0, 4, 1)])
    with pytest.raises(tfhe_amd.GswError, match="canonical"):
        gsw.eval_circuit(pool, [("mult_const", 0, 2147483647, 1)])
    with pytest.raises(ValueError, match="unknown gate"):
        gsw.eval_circuit(pool, [("xor", 0, 1, 2)])
    gsw.eval_circuit(pool, [])


@pytest.mark.parametrize("ring_name", ["fp", "z30"])
def test_partial_decryptions_of_additive_shares_recombine(oracle, tfhe_amd, ring_name):
    """k parties each hold an additive share of sk.s; the sum of their partial decryptions is the value
    decrypt tests (gsw_impl.rs:48-52), so the recombined bits equal plain decryption."""
    ring_d = tfhe_amd.Fp() if ring_name == "fp" else tfhe_amd.Zpow2(30)
    gsw = tfhe_amd.GSW(ring_d, 10, 10 * ring_d.ell, seed=31)
    sk, pk = gsw.keygen()
    msgs = [0, 1, 1, 0, 1]
    ct = gsw.nand(gsw.encrypt_batch(pk, msgs), gsw.encrypt_batch(pk, [1] * 5))
    bits, raw = gsw.decrypt_batch(sk, ct, raw=True)
    assert bits.tolist() == [1 - m for m in msgs]
    for parties in (1, 2, 5):
        shares = gsw.share_secret(sk, parties, seed=parties)
        assert np.array_equal(shares.astype(np.uint64).sum(axis=0) % ring_d.q, sk.s)
        partials = np.stack([gsw.part_dec(shares[p], ct) for p in range(parties)])
        assert np.array_equal(partials.astype(np.uint64).sum(axis=0) % ring_d.q, raw)
        assert gsw.combine_partials(partials).tolist() == bits.tolist()
    with pytest.raises(ValueError):
        gsw.part_dec(np.zeros(3), ct)


def test_full_benchmark_batch_through_slot_lists(oracle, tfhe_amd):
    """1024 products of the n = 256 benchmark shape in one gate list over a pool of 16 operands: entry b
    multiplies operands (2b mod 16, 2b + 1 mod 16), so the digests must repeat with period 8 (a checksum of
    checksums over the whole batch), and two full products are compared with the oracle bit for bit."""
    from tfhe_amd import _lib

    L = _lib.lib()
    ring_o, ring_d = oracle.fp(), tfhe_amd.Fp()
    n, operands, batch = 256, 16, 1024
    gsw = tfhe_amd.GSW(ring_d, n, n * 31, seed=5)
    eng = gsw.engine
    pool = tfhe_amd.Ciphertext(eng, operands + batch)
    _lib.check(L.gsw_memset_zero(eng.handle, pool.ptr, 4 * eng.compact_elems * (operands + batch)))
    rng = np.random.default_rng(77)
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
    og = oracle.GSW(ring_o, n, n * 31)
    out = np.empty((eng.big_n, n + 1), dtype=np.uint32)
    for b in (3, 1020):
        _lib.check(L.gsw_download_compact(eng.handle, out.ctypes.data, pool.entry_ptr(operands + b)))
        k = b % 8
        assert np.array_equal(out, og.mul_gadget(host[2 * k], host[2 * k + 1])), b
