#!/bin/bash
# One SQ counter pass (MFMA busy, issued MFMA ops, clock, wave cycles and waits) of the contraction for one bench.py
# preset at launches of <batch> products, serial schedule (run through gpurun):
#   tools/profile_sq_shape.sh "<bench args>" <batch> <tag>
set -o pipefail
ARGS=$1; B=$2; TAG=$3
R=${GRAFT_REPO_ROOT:-$PWD}
OUT=$R/gpurun_out/prof_sq_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
export GSW_OVERLAP_CHUNK=0
timeout -s KILL 300 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_I8 SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE -d $OUT/sq -o p --output-format csv -- python3 $R/bench.py $ARGS --no-cpu-baseline --batch $B --pool $((2 * B)) --steps 2 --warmup 1 > $OUT/sq.json 2> $OUT/sq.err || { tail -5 $OUT/sq.err; exit 1; }
cd $R
python tools/pmc_summary.py gpurun_out/prof_sq_$TAG/sq gemm16 > gpurun_out/pmc_sq_$TAG.txt
cat gpurun_out/pmc_sq_$TAG.txt
