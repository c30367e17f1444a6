#!/bin/bash
# All profiling passes of one round, on the GPU box (run from the repository root through gpurun):
#   1. kernel-trace statistics of the default bench command
#   2. PMC passes, each in a run of its own: FETCH_SIZE, WRITE_SIZE (HBM-side traffic of the contraction),
#      MFMA busy / wave cycles / clock, LDS activity
#   3. kernel-trace statistics of the dense-product and gadget-decomposition timing tools, and a marker trace
#      (GSW_ROCTX=1) of the chain / encryption job
# Outputs land under gpurun_out/prof/; the summaries worth keeping are copied into profiles/ by hand.
set -o pipefail
R=${GRAFT_REPO_ROOT:-$PWD}
OUT=$R/gpurun_out/prof
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout -k 10 400 rocprofv3 --kernel-trace --stats -d $OUT/stats -o s --output-format csv -- python3 $R/bench.py --steps 6 --warmup 2 --cpu-seconds 5 > $OUT/bench_under_rocprof.json 2> $OUT/stats.err || { tail -5 $OUT/stats.err; exit 1; }
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $OUT/dense -o s --output-format csv -- python3 $R/tools/bench_dense.py 256 > $OUT/dense.jsonl 2> $OUT/dense.err || { tail -5 $OUT/dense.err; exit 1; }
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $OUT/decomp -o s --output-format csv -- python3 $R/tools/bench_decomp.py 256 > $OUT/decomp.jsonl 2> $OUT/decomp.err || { tail -5 $OUT/decomp.err; exit 1; }
GSW_ROCTX=1 GSW_EXTRA_CHAINS=32 timeout -k 10 300 rocprofv3 --kernel-trace --marker-trace --stats -d $OUT/markers -o s --output-format csv -- python3 $R/tools/bench_extra.py > $OUT/markers.json 2> $OUT/markers.err || { tail -5 $OUT/markers.err; exit 1; }
export GSW_OVERLAP_CHUNK=0
pmc() {  # name, counters...
  local name=$1; shift
  timeout -s KILL 200 rocprofv3 --kernel-trace --pmc "$@" -d $OUT/$name -o p --output-format csv -- python3 $R/bench.py --no-cpu-baseline --batch 512 --steps 2 --warmup 1 > $OUT/$name.json 2> $OUT/$name.err || { tail -5 $OUT/$name.err; exit 1; }
}
pmc fetch FETCH_SIZE && pmc write WRITE_SIZE && \
pmc sq SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_I8 SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE && \
pmc lds SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_INSTS_LDS SQ_INSTS_VALU SQ_INSTS_MFMA || exit 1
cd $R
for n in fetch write sq lds; do echo "== $n"; python tools/pmc_summary.py gpurun_out/prof/$n gemm16; done
cat $OUT/bench_under_rocprof.json
