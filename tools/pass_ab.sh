#!/bin/bash
# placement passes (row tiles x column tiles per XCD pass) A/B in one session, serial schedule
# usage: tools/pass_ab.sh "8x8 8x9 16x9 ..." [bench args]
SHAPES=$1; shift
mkdir -p gpurun_out
for i in 1 2; do
for s in $SHAPES; do
  GSW_PASS_ROWS=${s%x*} GSW_PASS_COLS=${s#*x} GSW_OVERLAP_CHUNK=0 timeout -k 10 200 python bench.py --no-cpu-baseline --steps 4 "$@" > gpurun_out/pa.json 2> gpurun_out/pa.err || { tail -5 gpurun_out/pa.err; exit 1; }
  python - "$s" <<'PY'
import json, sys
d = json.load(open("gpurun_out/pa.json")); r = d["roofline"]
print(f"passes {sys.argv[1]:6s} {d['value']:9.0f} mults/s  gemm {r['kernel_avg_ms']:7.3f} ms  frac {r['frac']:.4f}")
PY
done
done
