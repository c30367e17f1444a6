#!/bin/bash
# A/B of one environment switch of the library inside one GPU session (serial schedule unless bench args say otherwise)
# usage: tools/env_ab.sh VAR "value1 value2 ..." [rounds] [bench args...]
VAR=$1; VALUES=$2; R=${3:-3}; shift 3
mkdir -p gpurun_out
for i in $(seq 1 $R); do
  for v in $VALUES; do
    env $VAR=$v GSW_OVERLAP_CHUNK=${GSW_OVERLAP_CHUNK-0} timeout -k 10 200 python bench.py --no-cpu-baseline --steps 4 "$@" > gpurun_out/eab.json 2> gpurun_out/eab.err || { tail -5 gpurun_out/eab.err; exit 1; }
    python - "$VAR=$v" <<'PY'
import json, sys
d = json.load(open("gpurun_out/eab.json")); r = d["roofline"]
print(f"{sys.argv[1]:28s} {d['value']:9.0f} mults/s  step {d['ms_per_step']:7.2f} ms  gemm {r['kernel_avg_ms']:7.3f} ms  frac {r['frac']:.4f}  verified {d['config']['exchange_verified']}")
PY
  done
done
