#!/usr/bin/env python
"""Parity of one contraction kernel variant of one library build against the CPU oracle (experiments):
GSW_HIP_LIBRARY=build/libX.so GSW_GEMM_VARIANT=V python tools/check_variant.py
Random operands with extreme rows over the shapes of the parity tests plus one n = 256 product (sampled rows)."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import tfhe_amd  # noqa: E402
from oracle import gsw_oracle as O  # noqa: E402  (a checker script, not product code)


def main():
    shapes = [("fp", 10), ("fp", 1), ("fp", 3), ("fp", 7), ("fp", 32), ("fp", 39), ("fp", 40), ("fp", 71), ("z30", 10),
              ("z32", 10), ("z32", 33), ("z8", 5), ("z1", 40), ("fp", 8), ("fp", 16), ("fp", 17), ("z32", 16), ("fp", 64)]
    rings = {"fp": (O.fp(), tfhe_amd.Fp()), "z30": (O.zpow2(30), tfhe_amd.Zpow2(30)), "z32": (O.zpow2(32), tfhe_amd.Zpow2(32)),
             "z8": (O.zpow2(8), tfhe_amd.Zpow2(8)), "z1": (O.zpow2(1), tfhe_amd.Zpow2(1))}
    for name, n in shapes:
        ring_o, ring_d = rings[name]
        gsw = tfhe_amd.GSW(ring_d, n, n * ring_d.ell, seed=1)
        og = O.GSW(ring_o, n, n * ring_o.ell)
        rng = np.random.default_rng(100 + n)
        shape = (3, ring_o.big_n(n), n + 1)
        c1 = rng.integers(0, ring_o.q, size=shape, dtype=np.uint64).astype(np.uint32)
        c2 = rng.integers(0, ring_o.q, size=shape, dtype=np.uint64).astype(np.uint32)
        c1[2] = ring_o.q - 1
        c2[2] = ring_o.q - 1
        d1, d2 = tfhe_amd.Ciphertext.from_compact(gsw.engine, c1), tfhe_amd.Ciphertext.from_compact(gsw.engine, c2)
        out = tfhe_amd.Ciphertext(gsw.engine, 3)
        out.buf.upload(np.full(3 * gsw.engine.compact_elems, 0xDEADBEEF, dtype=np.uint32))
        from tfhe_amd import _lib
        _lib.check(_lib.lib().gsw_mul_gadget_batch(gsw.engine.handle, d1.ptr, d1.stride, d2.ptr, d2.stride, out.ptr, out.stride, 3, 0))
        raw = out.buf.download(np.uint32, 3 * gsw.engine.compact_elems).reshape(3, gsw.engine.big_n, gsw.engine.ld)
        assert (raw[:, :, n + 1:] == 0).all(), (name, n, "padding")
        nand = gsw.nand(d1, d2).to_compact()
        for b in range(3):
            assert np.array_equal(raw[b, :, :n + 1], og.mul_gadget(c1[b], c2[b])), (name, n, b)
            assert np.array_equal(nand[b], og.mul_gadget(c1[b], c2[b], flags=O.FLAG_NAND)), (name, n, b, "nand")
        print("ok", name, n, "variant", _lib.lib().gsw_ctx_gemm_variant(gsw.engine.handle), flush=True)
    ring_o, ring_d = rings["fp"]
    n = 256
    gsw = tfhe_amd.GSW(ring_d, n, n * 31, seed=1)
    og = O.GSW(ring_o, n, n * 31)
    rng = np.random.default_rng(5)
    c1 = rng.integers(0, ring_o.q, size=(ring_o.big_n(n), n + 1), dtype=np.uint64).astype(np.uint32)
    c2 = rng.integers(0, ring_o.q, size=(ring_o.big_n(n), n + 1), dtype=np.uint64).astype(np.uint32)
    got = gsw.mult(tfhe_amd.Ciphertext.from_compact(gsw.engine, c1), tfhe_amd.Ciphertext.from_compact(gsw.engine, c2)).to_compact()
    big_n = ring_o.big_n(n)
    for r in [0, 1, 15, 16, 127, 128, 255, 256, 511, 512, 4000, big_n - 288, big_n - 33, big_n - 32, big_n - 31, big_n - 1]:
        assert np.array_equal(got[r], og.mul_gadget_rows(c1, c2, r, 1, 0)[0]), ("n=256 row", r)
    print("ok fp 256 sampled rows", flush=True)


if __name__ == "__main__":
    main()
