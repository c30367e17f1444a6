"""Independent numpy restatement of the wire format documented in include/gsw_hip.h (test helper).

Layout, little endian: "GSW1", kind u8, ring kind u8, bits u8, 0 u8, n, m, rows, cols, count, 0 (u32 each),
then count*rows*cols canonical u32 values row-major, then the u64 position-weighted digest of the payload.
"""
import struct

import numpy as np

CIPHERTEXT, PUBLIC_KEY, SECRET_KEY = 1, 2, 3
RING_FP, RING_ZPOW2 = 0, 1
GOLDEN_RATIO = 0x9E3779B97F4A7C15
MASK = (1 << 64) - 1


def digest(values) -> int:
    """sum over i of (x_i + 1)(2 i + 1), times the odd constant, modulo 2^64."""
    acc = 0
    for i, x in enumerate(np.asarray(values, dtype=np.uint32).reshape(-1).tolist()):
        acc = (acc + (x + 1) * (2 * i + 1)) & MASK
    return (acc * GOLDEN_RATIO) & MASK


def write(kind, ring_kind, bits, n, m, values) -> bytes:
    values = np.asarray(values, dtype="<u4")
    if values.ndim == 2:
        values = values[None]
    count, rows, cols = values.shape
    head = b"GSW1" + bytes([kind, ring_kind, bits, 0]) + struct.pack("<6I", n, m, rows, cols, count, 0)
    return head + values.tobytes() + struct.pack("<Q", digest(values))


def read(data: bytes):
    assert data[:4] == b"GSW1"
    kind, ring_kind, bits, zero = data[4:8]
    n, m, rows, cols, count, zero2 = struct.unpack("<6I", data[8:32])
    assert zero == 0 and zero2 == 0
    size = 4 * count * rows * cols
    assert len(data) == 32 + size + 8
    values = np.frombuffer(data[32:32 + size], dtype="<u4").reshape(count, rows, cols)
    (dig,) = struct.unpack("<Q", data[32 + size:])
    assert dig == digest(values)
    return dict(kind=kind, ring_kind=ring_kind, bits=bits, n=n, m=m), values
