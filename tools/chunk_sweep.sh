#!/bin/bash
# whole-job rate of the default bench against the chunk of the preparation / contraction pipeline
mkdir -p gpurun_out
for i in 1 2; do
for c in ${CHUNKS:-64 96 128 192 256 0}; do
  GSW_OVERLAP_CHUNK=$c timeout -k 10 200 python bench.py --no-cpu-baseline --steps 4 "$@" > gpurun_out/cs.json 2> gpurun_out/cs.err || { tail -5 gpurun_out/cs.err; exit 1; }
  python - "$c" <<'PY'
import json, sys
d = json.load(open("gpurun_out/cs.json")); r = d["roofline"]
print(f"chunk {sys.argv[1]:4s} {d['value']:9.0f} mults/s  step {d['ms_per_step']:7.2f} ms  gemm {r['kernel_avg_ms']:7.3f} ms x {r['launches'] // 4} per step  frac {r['frac']:.4f}")
PY
done
done
