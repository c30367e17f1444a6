#!/bin/bash
# What the padding of the n = 256 shape costs (timing only, wrong results): GSW_ABL_SHAPE bit 0 drops the
# one-column tail tile, bit 1 the thin last row tile, bit 2 the last K stage.  Needs a -DGSW_DIAGNOSTIC_BUILDS library.
LIB=${1:-build/libgsw_diag.so}
mkdir -p gpurun_out
for m in 0 1 2 4 7 0 1 2 4 7; do
  GSW_ABL_SHAPE=$m GSW_BENCH_DIAGNOSTIC=1 GSW_OVERLAP_CHUNK=0 GSW_HIP_LIBRARY=$PWD/$LIB timeout -k 10 200 python bench.py --no-cpu-baseline --steps 4 > gpurun_out/sa.json 2> gpurun_out/sa.err || { tail -5 gpurun_out/sa.err; exit 1; }
  python - "$m" <<'PY'
import json, sys
d = json.load(open("gpurun_out/sa.json")); r = d["roofline"]
print(f"GSW_ABL_SHAPE={sys.argv[1]}  gemm {r['kernel_avg_ms']:8.3f} ms / {r['products_per_launch']:.0f} products")
PY
done
