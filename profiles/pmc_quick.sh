#!/bin/bash
# PMC passes for the conv kernels (run on the GPU box from the repo root):  bash profiles/pmc_quick.sh <tag>
# Counter passes are separate runs with --kernel-trace only (never combined with other trace domains).
set -u
TAG=${1:-r1}
OUT=$GRAFT_REPO_ROOT/gpurun_out/pmc_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
ARGS="$GRAFT_REPO_ROOT/bench.py --steps 1 --warmup 0 --denoise-steps 10 --no-cpu-baseline --no-profile"
timeout -k 10 300 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU_MFMA_MOPS_F32 GRBM_GUI_ACTIVE --output-format csv -d $OUT/sq -- python3 $ARGS > $OUT/sq.log 2>&1
timeout -k 10 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/fetch -- python3 $ARGS > $OUT/fetch.log 2>&1
timeout -k 10 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --output-format csv -d $OUT/write -- python3 $ARGS > $OUT/write.log 2>&1
ls -R $OUT | head -30
