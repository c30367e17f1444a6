"""Timings of the operations either side of the product, at the n = 256 benchmark shape, with their
results checked by decryption: batched encryption (the same contraction with the random bit matrix as
the binary operand), depth-8 multiplication chains over a resident pool (gate lists), batched decryption.
Prints one JSON object.  Needs a GPU; run from the repository root.
"""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import tfhe_amd  # noqa: E402


def timed(engine, fn, repeat=3):
    fn()
    engine.sync()
    t0 = time.perf_counter()
    for _ in range(repeat):
        fn()
    engine.sync()
    return (time.perf_counter() - t0) / repeat


def main():
    n = int(os.environ.get("GSW_EXTRA_N", "256"))
    chains = int(os.environ.get("GSW_EXTRA_CHAINS", "128"))
    depth = 8
    ring = tfhe_amd.Fp()
    gsw = tfhe_amd.GSW(ring, n, n * ring.ell, seed=0x6A5C1000)
    eng = gsw.engine
    sk, pk = gsw.keygen()
    out = {"n": n, "N": eng.big_n, "m": n * ring.ell, "ring": "Fp 2^31-1"}

    # Pool: accumulators in [0, chains), their ping-pong partners in [chains, 2 chains) (a product may not
    # overwrite its operand), then one range of fresh factors per level.
    pool = tfhe_amd.Ciphertext(eng, chains * (2 + depth))
    ones = [1] * chains

    sec = timed(eng, lambda: gsw.encrypt_batch(pk, ones, out=pool, slot=0))
    out["encrypt"] = {"batch": chains, "encryptions_per_s": chains / sec,
                      "includes": "device generation of the N x m random bits, the contraction, message * G"}
    assert gsw.decrypt_batch(sk, pool, slot=0, count=chains).tolist() == ones

    zero_level, zero_chain = 3, chains // 2  # one chain gets a 0 factor at level 3
    for lvl in range(depth):
        msgs = [0 if (lvl == zero_level and c == zero_chain) else 1 for c in range(chains)]
        gsw.encrypt_batch(pk, msgs, out=pool, slot=chains * (2 + lvl))
    gates = []
    for lvl in range(depth):
        src, dst = (0, chains) if lvl % 2 == 0 else (chains, 0)
        gates += [("mult", src + c, chains * (2 + lvl) + c, dst + c) for c in range(chains)]
    final = 0 if depth % 2 == 0 else chains
    gates = gsw.encode_gates(gates)  # the C ABI's gate array, made once: the timed runs hand it over as it is

    gsw.eval_circuit(pool, gates)  # acc <- mult(acc, fresh_level), level after level
    eng.sync()
    bits = gsw.decrypt_batch(sk, pool, slot=final, count=chains).tolist()
    want = [0 if c == zero_chain else 1 for c in range(chains)]
    assert bits == want, "chain results do not decrypt to the product of their factors"
    # timing: running the list again multiplies the results by the same factors once more -- the same work
    sec = timed(eng, lambda: gsw.eval_circuit(pool, gates), repeat=2)
    out["chains_depth8"] = {"chains": chains, "products": chains * depth, "seconds_per_job": sec,
                            "products_per_s": chains * depth / sec, "launch_sequences": depth,
                            "decrypts_correctly": True}

    # the same chains with the fresh factors already in prepared form (encrypt_batch(..., cache=)): no preparation
    # launch at any level, one contraction launch per level
    cache = tfhe_amd.PreparedCache(eng, pool.batch)
    gsw.encrypt_batch(pk, ones, out=pool, slot=0)
    for lvl in range(depth):
        msgs = [0 if (lvl == zero_level and c == zero_chain) else 1 for c in range(chains)]
        gsw.encrypt_batch(pk, msgs, out=pool, slot=chains * (2 + lvl), cache=cache)
    prepared0, _ = eng.counters()
    gsw.eval_circuit(pool, gates, cache=cache)
    eng.sync()
    assert gsw.decrypt_batch(sk, pool, slot=final, count=chains).tolist() == want
    sec = timed(eng, lambda: gsw.eval_circuit(pool, gates, cache=cache), repeat=2)
    out["chains_depth8_prepared_factors"] = {"chains": chains, "products": chains * depth, "seconds_per_job": sec,
                                             "products_per_s": chains * depth / sec,
                                             "operands_prepared_during_the_chains": eng.counters()[0] - prepared0,
                                             "decrypts_correctly": True}

    sec = timed(eng, lambda: gsw.decrypt_batch(sk, pool, slot=final, count=chains))
    out["decrypt"] = {"batch": chains, "decryptions_per_s": chains / sec}
    print(json.dumps(out))


if __name__ == "__main__":
    main()
