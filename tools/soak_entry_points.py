"""Determinism soak of the other entry points that run the contraction: the entry point that allocates nothing, and
gate lists over a pool (with and without the prepared-operand cache).  The same work repeated must give the same
64-bit digests every time.  Needs a GPU.  usage: soak_entry_points.py [ring: fp|z32] [n] [products] [repeats]"""
import ctypes as C
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import tfhe_amd  # noqa: E402
from tfhe_amd import _lib  # noqa: E402


def digests(eng, L, ptr, ct, count):
    out = torch.zeros(count, dtype=torch.int64, device="cuda:0")
    torch.cuda.synchronize()
    _lib.check(L.gsw_digest_batch(eng.handle, ptr, ct, ct, count, out.data_ptr()))
    eng.sync()
    return out.cpu().numpy().copy()


def main():
    ring_name = sys.argv[1] if len(sys.argv) > 1 else "fp"
    n = int(sys.argv[2]) if len(sys.argv) > 2 else 10
    count = int(sys.argv[3]) if len(sys.argv) > 3 else 1024
    repeats = int(sys.argv[4]) if len(sys.argv) > 4 else 50
    ring = tfhe_amd.Fp() if ring_name == "fp" else tfhe_amd.Zpow2(32)
    gsw = tfhe_amd.GSW(ring, n, n * ring.ell, seed=7, check_max=False)
    eng, L = gsw.engine, _lib.lib()
    ct = eng.compact_elems
    pool = tfhe_amd.Ciphertext(eng, 3 * count)  # slots 0 .. 2 count - 1: operands, 2 count ..: results
    _lib.check(L.gsw_fill_uniform(eng.handle, 11, 0, 0, ring.max_u64(), 1, pool.ptr, 3 * count * ct))
    pool = tfhe_amd.Ciphertext.from_compact(eng, pool.to_compact())  # the same values with the padding columns zeroed
    res_ptr = pool.entry_ptr(2 * count)
    failures = 0

    def check(label, rep, d, first):
        nonlocal failures
        if first is not None and not np.array_equal(d, first):
            bad = np.nonzero(d != first)[0]
            print(f"{label}, run {rep}: {len(bad)} of {count} products differ, first entries {bad[:8].tolist()}", flush=True)
            failures += 1

    # the entry point that allocates nothing against the allocating one
    _lib.check(L.gsw_mul_gadget_batch(eng.handle, pool.ptr, 2 * ct, pool.ptr + 4 * ct, 2 * ct, res_ptr, ct, count, 0))
    want = digests(eng, L, res_ptr, ct, count)
    for rep in range(repeats):
        _lib.check(L.gsw_mul_gadget_batch_static(eng.handle, pool.ptr, 2 * ct, pool.ptr + 4 * ct, 2 * ct, res_ptr, ct, count, 0))
        check("static entry point", rep, digests(eng, L, res_ptr, ct, count), want)
    print("static entry point done", flush=True)
    # one level of nand gates over the pool, again and again: (2 i, 2 i + 1) -> 2 count + i
    gates = gsw.encode_gates([("nand", 2 * i, 2 * i + 1, 2 * count + i) for i in range(count)])
    first = None
    for rep in range(repeats):
        gsw.eval_circuit(pool, gates)
        d = digests(eng, L, res_ptr, ct, count)
        check("gate list", rep, d, first)
        first = d if first is None else first
    print("gate list done", flush=True)
    cache = tfhe_amd.PreparedCache(eng, 3 * count)
    for rep in range(repeats):
        gsw.eval_circuit(pool, gates, cache=cache)
        check("gate list with the prepared cache", rep, digests(eng, L, res_ptr, ct, count), first)
    print("cached gate list done", flush=True)
    if failures:
        raise SystemExit(f"soak FAILED: {failures} runs differ")
    print(f"soak ok: {ring_name} n={n} products={count} x{repeats} (static entry point, gate lists, cached gate lists)")


if __name__ == "__main__":
    main()
