#!/bin/bash
# Time per tile against the K depth of a tile (32 / 64 / 128 stages), on shapes with no padding: the intercept of
# the line is what a tile costs beside its stages (launch, pipeline fill, epilogue).  Serial schedule.
mkdir -p gpurun_out
for w in fp_n127 fp_n255 fp_n511 fp_n127 fp_n255 fp_n511; do
  GSW_OVERLAP_CHUNK=0 timeout -k 10 200 python bench.py --no-cpu-baseline --steps 4 --workload $w > gpurun_out/to.json 2> gpurun_out/to.err || { tail -5 gpurun_out/to.err; exit 1; }
  python - "$w" <<'PY'
import json, sys
d = json.load(open("gpurun_out/to.json")); r = d["roofline"]
n = int(sys.argv[1][4:]); rows = 31 * (n + 1); tiles = (n + 1) // 32 * ((rows + 255) // 256); stages = (n + 1) // 4
per_tile_us = r["kernel_avg_ms"] * 1e3 / (r["products_per_launch"] * tiles) * 512  # 512 workgroup slots on the chip
print(f"{sys.argv[1]:8s} stages {stages:4d} tiles/product {tiles:5d} gemm {r['kernel_avg_ms']:8.3f} ms / {r['products_per_launch']:.0f} products  frac {r['frac']:.4f}  {per_tile_us:8.2f} us per tile  {per_tile_us / stages:6.3f} us per stage")
PY
done
