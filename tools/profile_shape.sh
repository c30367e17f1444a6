#!/bin/bash
# FETCH_SIZE / WRITE_SIZE passes and kernel statistics for one workload of bench.py (run through gpurun):
#   tools/profile_shape.sh <workload> <batch>
set -o pipefail
W=$1; B=$2
R=${GRAFT_REPO_ROOT:-$PWD}
OUT=$R/gpurun_out/prof_$W
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout -k 10 400 rocprofv3 --kernel-trace --stats -d $OUT/stats -o s --output-format csv -- python3 $R/bench.py --workload $W --steps 3 --warmup 1 --no-cpu-baseline > $OUT/bench_under_rocprof.json 2> $OUT/stats.err || { tail -5 $OUT/stats.err; exit 1; }
export GSW_OVERLAP_CHUNK=0
for c in FETCH_SIZE WRITE_SIZE; do
  timeout -s KILL 300 rocprofv3 --kernel-trace --pmc $c -d $OUT/$c -o p --output-format csv -- python3 $R/bench.py --workload $W --no-cpu-baseline --batch $B --steps 2 --warmup 1 > $OUT/$c.json 2> $OUT/$c.err || { tail -5 $OUT/$c.err; exit 1; }
done
cd $R
for c in FETCH_SIZE WRITE_SIZE; do python tools/pmc_summary.py gpurun_out/prof_$W/$c gemm16; done
head -3 $OUT/stats/s_kernel_stats.csv | cut -c1-170
