export GSW_HIP_LIBRARY=$PWD/build/libgsw_persist.so
echo "== n=256 (4 launches of 256 products, serial schedule)"
bash tools/env_ab.sh GSW_PERSIST "0 1" 3 || exit 1
echo "== n=512 Fp (launches of 128 products)"
bash tools/env_ab.sh GSW_PERSIST "0 1" 2 --workload fp_n512 --batch 128 || exit 1
echo "== bit-exact check of the persistent grid against the CPU port (n=256, 32 products)"
GSW_PERSIST=1 GSW_OVERLAP_CHUNK=0 timeout -k 10 200 python bench.py --batch 64 --steps 1 --warmup 0 --cpu-seconds 3 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('gpu_matches_cpu_bit_exact', d['cpu_baseline']['gpu_matches_cpu_bit_exact'], 'exchange_verified', d['config']['exchange_verified'])" || exit 1
cd /tmp && export TMPDIR=/tmp GSW_OVERLAP_CHUNK=0
for pz in 0 1; do for w in fp_n256:256 fp_n512:128; do wl=${w%%:*}; b=${w##*:}
  GSW_PERSIST=$pz timeout -s KILL 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d /tmp/pf_${pz}_$wl -o p --output-format csv -- python3 $GRAFT_REPO_ROOT/bench.py --workload $wl --no-cpu-baseline --batch $b --steps 2 --warmup 1 > /dev/null 2>/tmp/pf.err || { tail -3 /tmp/pf.err; exit 1; }
  echo "== FETCH_SIZE GSW_PERSIST=$pz $wl batch $b"; python3 $GRAFT_REPO_ROOT/tools/pmc_summary.py /tmp/pf_${pz}_$wl gemm16
done; done
