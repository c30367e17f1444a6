#!/bin/bash
# Experiment: persistent grid that takes tiles off a per-XCD counter (GSW_PERSIST=2, -DGSW_EXP_PERSIST library) against
# the hardware dispatcher (GSW_PERSIST=0), serial schedule
export GSW_HIP_LIBRARY=$PWD/build/libgsw_persist.so
echo "== bit-exact check against the CPU port (n=256, 64 products)"
GSW_PERSIST=${PERSIST_CHECK:-2} GSW_OVERLAP_CHUNK=0 timeout -k 10 200 python bench.py --batch 64 --steps 1 --warmup 0 --cpu-seconds 3 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('gpu_matches_cpu_bit_exact', d['cpu_baseline']['gpu_matches_cpu_bit_exact'], 'exchange_verified', d['config']['exchange_verified'])" || exit 1
echo "== n=256 (4 launches of 256 products)"
bash tools/env_ab.sh GSW_PERSIST "${PERSIST_MODES:-0 2 1}" 3 || exit 1
echo "== n=512 Fp (launches of 128 products)"
bash tools/env_ab.sh GSW_PERSIST "${PERSIST_MODES:-0 2 1}" 2 --workload fp_n512 --batch 128 || exit 1
