#!/bin/bash
# PMC passes for the conv kernels on the headline workload (BASELINE configs[2], 10 denoising steps): FETCH_SIZE and
# WRITE_SIZE + LDS counters in separate passes -> gpurun_out/<tag>/pmc_{fetch,write}; profiles/summarize_r02.py reads
# gpurun_out/r02/, so pass tag r02 to refresh profiles/r02/traffic.json:   bash profiles/pmc_conv.sh r02
set -u
TAG=${1:-r02}
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
ARGS="$R/bench.py --steps 1 --warmup 0 --denoise-steps 10 --no-cpu-baseline --no-profile --no-extras"
rm -rf $OUT/pmc_fetch $OUT/pmc_write $OUT/pmc_sq
timeout -k 10 300 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F32 GRBM_GUI_ACTIVE --output-format csv -d $OUT/pmc_sq -o p -- python3 $ARGS > $OUT/pmc_sq.log 2>&1 && \
timeout -k 10 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -o p -- python3 $ARGS > $OUT/pmc_fetch.log 2>&1 && \
timeout -k 10 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --output-format csv -d $OUT/pmc_write -o p -- python3 $ARGS > $OUT/pmc_write.log 2>&1
find $OUT -name '*_kernel_trace.csv' -size +8M -delete
ls $OUT/pmc_fetch $OUT/pmc_write | head
