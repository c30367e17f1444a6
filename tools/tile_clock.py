"""Diagnostic (needs a -DGSW_EXP_TILE_CLOCK build named by GSW_HIP_LIBRARY): mean shader-clock cycles per workgroup of
each kind in the contraction -- full tile, last column tile, last row tile, both -- at n = 256 (or --n)."""
import argparse
import ctypes as C
import os
import sys

import numpy as np
import torch  # noqa: F401

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import tfhe_amd  # noqa: E402
from tfhe_amd import _lib  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--n", type=int, default=256)
ap.add_argument("--batch", type=int, default=256)
ap.add_argument("--ring", default="fp")
args = ap.parse_args()
ring = tfhe_amd.Fp() if args.ring == "fp" else tfhe_amd.Zpow2(32)
gsw = tfhe_amd.GSW(ring, args.n, args.n * ring.ell, seed=1, check_max=False)
eng, L = gsw.engine, _lib.lib()
L.gsw_diag_tile_clock.restype = C.c_int
L.gsw_diag_tile_clock.argtypes = [C.c_void_p, C.c_void_p, C.c_int]
ct = eng.compact_elems
pool = tfhe_amd.Ciphertext(eng, 2 * args.batch)
_lib.check(L.gsw_fill_uniform(eng.handle, 5, 1, 0, ring.max_u64(), 1, pool.ptr, 2 * args.batch * ct))
prep = eng.alloc(eng.prepared_bytes * args.batch)
out = tfhe_amd.Ciphertext(eng, args.batch)
_lib.check(L.gsw_prepare_rhs(eng.handle, pool.entry_ptr(args.batch), ct, prep.ptr, args.batch))
res = (C.c_uint64 * 16)()
_lib.check(L.gsw_diag_tile_clock(eng.handle, res, 1))  # reset (earliest start = max)
import time  # noqa: E402
for rep in range(3):
    t0 = time.perf_counter()
    _lib.check(L.gsw_mul_gadget_prepared(eng.handle, pool.ptr, ct, prep.ptr, eng.prepared_bytes, out.ptr, ct, args.batch, 0))
    _lib.check(L.gsw_diag_tile_clock(eng.handle, res, 1))  # waits for the stream
    wall_us = (time.perf_counter() - t0) * 1e6
names = ["full tile", "last column tile", "last row tile", "last row + last column"]
base = res[0] / max(res[1], 1)
for k in range(4):
    if res[2 * k + 1]:
        mean = res[2 * k] / res[2 * k + 1]
        print(f"{names[k]:26s} {res[2 * k + 1]:8d} workgroups, mean {mean:10.0f} cycles = {mean / base:5.2f} full tiles, "
              f"share of workgroup time {res[2 * k] / sum(res[0:8:2]):6.3f}")
span = res[9] - res[8]
slots = 512  # 256 CUs x 2 resident workgroups
print(f"host wall time of the call {wall_us:.1f} us; launch span {span / 100:.1f} us (100 MHz counter), workgroup time {res[10] / 100:.1f} us over {slots} slots: "
      f"occupied share {res[10] / (slots * span):.4f}")
