"""Determinism soak: the same batch of products run repeatedly must give identical digests every time
(an intermittent race in the pipelined kernels would show as a digest that changes between runs), and a
sample of the products is compared with a plain second computation through the dense parity kernel's
independent path (BitDecomp + exact dim^3 product) at small n.  Needs a GPU.
usage: soak.py [ring: fp|z32] [n] [batch] [repeats]"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import tfhe_amd  # noqa: E402
from tfhe_amd import _lib  # noqa: E402


def main():
    ring_name = sys.argv[1] if len(sys.argv) > 1 else "fp"
    n = int(sys.argv[2]) if len(sys.argv) > 2 else 256
    batch = int(sys.argv[3]) if len(sys.argv) > 3 else 1024
    repeats = int(sys.argv[4]) if len(sys.argv) > 4 else 10
    ring = tfhe_amd.Fp() if ring_name == "fp" else tfhe_amd.Zpow2(32)
    gsw = tfhe_amd.GSW(ring, n, n * ring.ell, seed=99)
    eng, L = gsw.engine, _lib.lib()
    dev = torch.device("cuda", 0)
    ct = eng.compact_elems
    pool = torch.empty((2 * batch, eng.big_n, eng.ld), dtype=torch.int32, device=dev)
    _lib.check(L.gsw_fill_uniform(eng.handle, 99, 0, 0, ring.max_u64(), 1, pool.data_ptr(), 2 * batch * ct))
    pool[:, :, n + 1:] = 0
    out = torch.empty((batch, eng.big_n, eng.ld), dtype=torch.int32, device=dev)
    dig = torch.zeros(batch, dtype=torch.int64, device=dev)
    torch.cuda.synchronize()
    _lib.check(L.gsw_ctx_set_workspace_limit(eng.handle, batch * eng.prepared_bytes))
    first = None
    failures = 0
    for rep in range(repeats):
        flags = rep & 1  # alternate mult / nand so that consecutive runs differ in the epilogue
        _lib.check(L.gsw_mul_gadget_batch_digest(eng.handle, pool.data_ptr(), 2 * ct, pool.data_ptr() + 4 * ct, 2 * ct,
                                                 out.data_ptr(), ct, batch, flags, dig.data_ptr()))
        eng.sync()
        d = dig.cpu().numpy().copy()
        check = torch.zeros(batch, dtype=torch.int64, device=dev)
        torch.cuda.synchronize()  # torch fills on its own stream; the engine works on another
        _lib.check(L.gsw_digest_batch(eng.handle, out.data_ptr(), ct, ct, batch, check.data_ptr()))
        eng.sync()
        assert np.array_equal(d, check.cpu().numpy()), f"run {rep}: fused digests differ from digests of the stored products"
        if rep < 2:
            first = (first or {})
            first[flags] = d
        elif not np.array_equal(d, first[flags]):
            bad = np.nonzero(d != first[flags])[0]
            msg = f"run {rep}: digests of {len(bad)} of {batch} products changed between identical runs, first entries {bad[:8].tolist()}"
            if not os.environ.get("SOAK_KEEP_GOING"):
                raise AssertionError(msg)
            print(msg, flush=True)
            failures += 1
            continue
        print(f"run {rep} ok", flush=True)
    if failures:
        raise SystemExit(f"soak FAILED: {failures} of {repeats} runs differ")
    print(f"soak ok: {ring_name} n={n} batch={batch} x{repeats}")


if __name__ == "__main__":
    main()
