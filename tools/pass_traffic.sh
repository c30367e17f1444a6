#!/bin/bash
# FETCH_SIZE (L2-miss read traffic) and duration of the contraction for several placement-pass shapes of one workload:
#   tools/pass_traffic.sh <workload> <batch> "<rows>x<cols> ..."
set -o pipefail
W=$1; B=$2; SHAPES=$3
R=${GRAFT_REPO_ROOT:-$PWD}
OUT=$R/gpurun_out/pass_traffic_$W
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
export GSW_OVERLAP_CHUNK=0
for s in $SHAPES; do
  export GSW_PASS_ROWS=${s%x*} GSW_PASS_COLS=${s#*x}
  timeout -s KILL 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $OUT/$s -o p --output-format csv -- python3 $R/bench.py --workload $W --no-cpu-baseline --batch $B --steps 2 --warmup 1 > $OUT/$s.json 2> $OUT/$s.err || { tail -5 $OUT/$s.err; exit 1; }
  echo "== $W passes $s"; (cd $R && python tools/pmc_summary.py gpurun_out/pass_traffic_$W/$s gemm16)
done
