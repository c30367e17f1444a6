#!/bin/bash
# SQ counter pass (MFMA busy cycles, clock) of one library build / kernel variant:  tools/pmc_variant.sh <lib.so> <variant> <tag>
set -o pipefail
R=${GRAFT_REPO_ROOT:-$PWD}
OUT=$R/gpurun_out/pmcv_$3
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
export GSW_OVERLAP_CHUNK=0 GSW_HIP_LIBRARY=$R/$1
timeout -s KILL 200 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_I8 SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE -d $OUT -o p --output-format csv -- python3 $R/bench.py --no-cpu-baseline --batch 512 --steps 2 --warmup 1 --variant $2 > $OUT/bench.json 2> $OUT/err.log || { tail -5 $OUT/err.log; exit 1; }
cd $R
python tools/pmc_summary.py gpurun_out/pmcv_$3 gemm16
