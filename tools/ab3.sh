#!/bin/bash
# A/B/C timing of several builds of the library in one GPU session (interleaved rounds, serial schedule).
# usage: tools/ab3.sh "<libA.so> <libB.so> ..." [rounds] [extra bench args...]
LIBS=$1; R=${2:-3}; shift 2
mkdir -p gpurun_out
for i in $(seq 1 $R); do
  for l in $LIBS; do
    GSW_OVERLAP_CHUNK=0 GSW_HIP_LIBRARY=$PWD/$l timeout -k 10 200 python bench.py --no-cpu-baseline --steps 4 "$@" > gpurun_out/ab.json 2> gpurun_out/ab.err || { tail -5 gpurun_out/ab.err; exit 1; }
    python - "$l" <<'PY'
import json, sys
d = json.load(open("gpurun_out/ab.json")); r = d["roofline"]
print(f"{sys.argv[1]:40s} {d['value']:9.0f} mults/s  step {d['ms_per_step']:7.2f} ms  gemm {r['kernel_avg_ms']:7.3f} ms  frac {r['frac']:.4f}")
PY
  done
done
