#!/bin/bash
# Experiment: eight-wave 512-row tiles (variant 12 of -DGSW_EXP_WIDE8 libraries, ring depth 5 and 3) against the default kernel
run() { # label, env...
  local label=$1; shift
  env "$@" GSW_OVERLAP_CHUNK=0 timeout -k 10 200 python bench.py --no-cpu-baseline --steps 4 $BENCH_ARGS > gpurun_out/w8.json 2> gpurun_out/w8.err || { tail -5 gpurun_out/w8.err; exit 1; }
  python - "$label" <<'PY'
import json, sys
d = json.load(open("gpurun_out/w8.json")); r = d["roofline"]
print(f"{sys.argv[1]:34s} {d['value']:9.0f} mults/s  gemm {r['kernel_avg_ms']:7.3f} ms  frac {r['frac']:.4f}  verified {d['config']['exchange_verified']}")
PY
}
mkdir -p gpurun_out
for lib in build/libgsw_wide8_d5.so build/libgsw_wide8_d3.so; do
  echo "== parity of variant 12 in $lib"
  GSW_HIP_LIBRARY=$PWD/$lib GSW_GEMM_VARIANT=12 timeout -k 10 300 python tools/check_variant.py || exit 1
done
for BENCH_ARGS in "" "--workload fp_n512 --batch 128"; do
  echo "== bench args: '$BENCH_ARGS'"
  for i in 1 2; do
    run "default" GSW_HIP_LIBRARY=$PWD/tfhe_amd/libgsw_hip.so || exit 1
    for pr in 4 8; do for pc in 8 4; do
      run "wide8 depth5 pass ${pr}x${pc}" GSW_HIP_LIBRARY=$PWD/build/libgsw_wide8_d5.so GSW_GEMM_VARIANT=12 GSW_PASS_ROWS=$pr GSW_PASS_COLS=$pc || exit 1
    done; done
    run "wide8 depth3 pass 4x8" GSW_HIP_LIBRARY=$PWD/build/libgsw_wide8_d3.so GSW_GEMM_VARIANT=12 GSW_PASS_ROWS=4 GSW_PASS_COLS=8 || exit 1
  done
done
