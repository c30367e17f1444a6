"""Writes tests/golden/wire_*.bin: wire objects (format of include/gsw_hip.h) built by the numpy
restatement in tests/wire_format.py from seeded data.  Run from the repository root:
    python tests/golden/make_wire_golden.py
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
import wire_format as wf  # noqa: E402

P = 2147483647


def payloads():
    rng = np.random.default_rng(20251111)
    n, m = 2, 62
    yield "wire_ct_fp_n2.bin", wf.write(wf.CIPHERTEXT, wf.RING_FP, 31, n, m,
                                        rng.integers(0, P, size=(3, 31 * (n + 1), n + 1), dtype=np.uint64))
    yield "wire_pk_fp_n2.bin", wf.write(wf.PUBLIC_KEY, wf.RING_FP, 31, n, m,
                                        rng.integers(0, P, size=(m, n + 1), dtype=np.uint64))
    yield "wire_sk_fp_n2.bin", wf.write(wf.SECRET_KEY, wf.RING_FP, 31, n, m,
                                        rng.integers(0, P, size=(1, n), dtype=np.uint64))
    n, m = 3, 96
    yield "wire_ct_z32_n3.bin", wf.write(wf.CIPHERTEXT, wf.RING_ZPOW2, 32, n, m,
                                         rng.integers(0, 1 << 32, size=(2, 32 * (n + 1), n + 1), dtype=np.uint64))


if __name__ == "__main__":
    for name, data in payloads():
        with open(os.path.join(HERE, name), "wb") as f:
            f.write(data)
        print(name, len(data))
