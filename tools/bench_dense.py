#!/usr/bin/env python
"""Times FheScheme::mult on the reference's own N x N ciphertext format (gsw_mul_dense): the MFMA path taken for
flattened left operands against the plain N^3 kernel, device-resident operands, HIP-event timing through
torch.  Prints one JSON line per size.  Usage: python tools/bench_dense.py [n ...]   (default 10 256)"""
import ctypes as C
import json
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import tfhe_amd  # noqa: E402
from tfhe_amd import _lib  # noqa: E402


def main():
    sizes = [int(a) for a in sys.argv[1:]] or [10, 256]
    L = _lib.lib()
    dev = torch.device("cuda", 0)
    for n in sizes:
        ring = tfhe_amd.Fp()
        gsw = tfhe_amd.GSW(ring, n, n * 31, seed=1)
        eng = gsw.engine
        stream = torch.cuda.Stream(dev)
        eng.set_stream(stream.cuda_stream)
        big_n = eng.big_n
        with torch.cuda.stream(stream):
            g = torch.Generator(device=dev).manual_seed(n)
            c1 = torch.randint(0, 2, (big_n, big_n), dtype=torch.int32, device=dev, generator=g)  # flattened: 0 / 1
            c1 += torch.eye(big_n, dtype=torch.int32, device=dev)                                 # + mu * I, mu = 1
            c2 = torch.randint(0, 2147483647, (big_n, big_n), dtype=torch.int32, device=dev, generator=g)
            out = torch.empty_like(c1)
            ref = torch.empty_like(c1)
            result = {"n": n, "N": big_n, "ring": "Fp"}
            for name, fast, reps in (("mfma", 2, 5), ("plain", 0, 1 if n > 64 else 5)):
                _lib.check(L.gsw_ctx_set_dense_fast(eng.handle, fast))
                dst = out if fast else ref
                _lib.check(L.gsw_mul_dense(eng.handle, c1.data_ptr(), c2.data_ptr(), dst.data_ptr(), big_n))  # warm-up
                assert L.gsw_ctx_last_dense_path(eng.handle) == (1 if fast else 0)
                start, stop = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                start.record(stream)
                for _ in range(reps):
                    _lib.check(L.gsw_mul_dense(eng.handle, c1.data_ptr(), c2.data_ptr(), dst.data_ptr(), big_n))
                stop.record(stream)
                stop.synchronize()
                ms = start.elapsed_time(stop) / reps
                result[f"{name}_ms"] = ms
                result[f"{name}_tops_int8_equiv"] = 2.0 * big_n ** 3 * 4 / (ms * 1e-3) / 1e12
            result["identical"] = bool(torch.equal(out, ref))
            result["speedup"] = result["plain_ms"] / result["mfma_ms"]
            # an arbitrary left operand (what mult(mult(a, b), c) multiplies by without flatten): sixteen signed-digit
            # int8 products on MFMA against the plain kernel (whose result `ref2` is computed once)
            c1.copy_(torch.randint(0, 2147483647, (big_n, big_n), dtype=torch.int32, device=dev, generator=g))
            for name, fast, reps in (("digits", 2, 3), ("plain_arbitrary", 0, 1 if n > 64 else 3)):
                _lib.check(L.gsw_ctx_set_dense_fast(eng.handle, fast))
                dst = out if fast else ref
                _lib.check(L.gsw_mul_dense(eng.handle, c1.data_ptr(), c2.data_ptr(), dst.data_ptr(), big_n))
                assert L.gsw_ctx_last_dense_path(eng.handle) == (2 if fast else 0)
                start, stop = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                start.record(stream)
                for _ in range(reps):
                    _lib.check(L.gsw_mul_dense(eng.handle, c1.data_ptr(), c2.data_ptr(), dst.data_ptr(), big_n))
                stop.record(stream)
                stop.synchronize()
                result[f"{name}_ms"] = start.elapsed_time(stop) / reps
            result["digits_tops_int8"] = 2.0 * big_n ** 3 * 16 / (result["digits_ms"] * 1e-3) / 1e12
            result["digits_identical"] = bool(torch.equal(out, ref))
            result["digits_speedup"] = result["plain_arbitrary_ms"] / result["digits_ms"]
        print(json.dumps(result), flush=True)


if __name__ == "__main__":
    main()
