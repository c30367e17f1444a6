"""Run under `rocprofv3 --pmc FETCH_SIZE`: streams known byte counts through the contraction's load
patterns so that the counter can be calibrated (see gsw_diag_stream_read in include/gsw_hip.h)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import tfhe_amd  # noqa: E402
from tfhe_amd import _lib  # noqa: E402

eng = tfhe_amd.Engine(tfhe_amd.Fp(), 256, 7936)
eng.set_stream(torch.cuda.current_stream().cuda_stream)
L = _lib.lib()
gib = 1 << 30
buf = torch.empty(4 * gib, dtype=torch.uint8, device="cuda")
buf.random_(0, 256)
torch.cuda.synchronize()
for _ in range(2):
    _lib.check(L.gsw_diag_stream_read(eng.handle, buf.data_ptr(), buf.numel(), 2 * gib, 0, 0))  # 1 KiB pieces
    # 128 MiB read as 64 B out of every 1040-B row (spans 2.03 GiB of the buffer)
    _lib.check(L.gsw_diag_stream_read(eng.handle, buf.data_ptr(), buf.numel(), gib // 8, 1, 1040))
torch.cuda.synchronize()
print("mode0 bytes", 2 * gib, "mode1 bytes", gib // 8)
