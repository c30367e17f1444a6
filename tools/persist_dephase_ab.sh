#!/bin/bash
# Experiment: in the dynamic persistent grid, the second workgroup of every CU starts a quarter / half tile late
run() { local label=$1; shift
  env "$@" GSW_OVERLAP_CHUNK=0 timeout -k 10 200 python bench.py --no-cpu-baseline --steps 4 > gpurun_out/pd.json 2> gpurun_out/pd.err || { tail -5 gpurun_out/pd.err; exit 1; }
  python - "$label" <<'PY'
import json, sys
d = json.load(open("gpurun_out/pd.json")); r = d["roofline"]
print(f"{sys.argv[1]:34s} {d['value']:9.0f} mults/s  gemm {r['kernel_avg_ms']:7.3f} ms  frac {r['frac']:.4f}  verified {d['config']['exchange_verified']}")
PY
}
for i in 1 2 3; do
  run "hardware dispatch" GSW_HIP_LIBRARY=$PWD/build/libgsw_persist.so GSW_PERSIST=0 || exit 1
  run "persistent, dynamic" GSW_HIP_LIBRARY=$PWD/build/libgsw_persist.so GSW_PERSIST=2 || exit 1
  run "persistent, dynamic, 37 k late" GSW_HIP_LIBRARY=$PWD/build/libgsw_persist_dp36.so GSW_PERSIST=2 || exit 1
  run "persistent, dynamic, 74 k late" GSW_HIP_LIBRARY=$PWD/build/libgsw_persist_dp72.so GSW_PERSIST=2 || exit 1
done
