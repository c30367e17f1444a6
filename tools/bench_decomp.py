#!/usr/bin/env python
"""Times the standalone gadget decomposition G^-1 (gsw_bit_decomp_as) of compact ciphertexts at the benchmark
shapes in its three output formats and reports the HBM rate (bytes read + bytes written over the kernel's time,
HIP events) against the 8 TB/s peak.  Usage: python tools/bench_decomp.py [n ...]   (default 256 512)"""
import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import tfhe_amd  # noqa: E402
from tfhe_amd import _lib  # noqa: E402

HBM_PEAK_GBS = 8000.0


def main():
    sizes = [int(a) for a in sys.argv[1:]] or [256, 512]
    L = _lib.lib()
    dev = torch.device("cuda", 0)
    for n in sizes:
        gsw = tfhe_amd.GSW(tfhe_amd.Fp(), n, n * 31, seed=1)
        eng = gsw.engine
        stream = torch.cuda.Stream(dev)
        eng.set_stream(stream.cuda_stream)
        with torch.cuda.stream(stream):
            batch = 8
            src = torch.empty((batch, eng.big_n, eng.ld), dtype=torch.int32, device=dev)
            _lib.check(L.gsw_fill_uniform(eng.handle, 3, 1, 0, 2147483646, 0, src.data_ptr(), batch * eng.compact_elems))
            for name, fmt, size in (("u32", _lib.DECOMP_U32, 4), ("u8", _lib.DECOMP_U8, 1), ("bits", _lib.DECOMP_BITS, 4)):
                per_row = L.gsw_bit_decomp_row_elems(eng.handle, n + 1, fmt)
                out_bytes = eng.big_n * per_row * size
                out = torch.empty(batch * out_bytes, dtype=torch.uint8, device=dev)
                reps = 5

                def run():
                    # the batch is contiguous, so its ciphertexts are simply more rows of one matrix: one launch
                    _lib.check(L.gsw_bit_decomp_as(eng.handle, src.data_ptr(), batch * eng.big_n, n + 1, eng.ld,
                                                   out.data_ptr(), fmt))
                run()
                start, stop = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                start.record(stream)
                for _ in range(reps):
                    run()
                stop.record(stream)
                stop.synchronize()
                ms = start.elapsed_time(stop) / (reps * batch)
                moved = eng.big_n * (n + 1) * 4 + out_bytes
                print(json.dumps({"n": n, "N": eng.big_n, "format": name, "ms_per_ciphertext": ms,
                                  "bytes_read": eng.big_n * (n + 1) * 4, "bytes_written": out_bytes,
                                  "gbs": moved / (ms * 1e-3) / 1e9, "frac_of_hbm_peak": moved / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS}),
                      flush=True)


if __name__ == "__main__":
    main()
