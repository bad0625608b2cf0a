#!/bin/bash
# Per-launch kernel trace of a few U-Net evaluations at the given batch sizes -> gpurun_out/<tag>/unet_B<agents>.txt
#   bash scripts/trace_unet.sh <tag> 1024 4096 [...]
set -u
TAG=$1; shift
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for B in "$@"; do
  rm -rf $OUT/tr_$B
  timeout -k 10 200 rocprofv3 --kernel-trace --output-format csv -d $OUT/tr_$B -o t -- python3 $R/scripts/one_unet.py $B > $OUT/tr_$B.log 2>&1 || { tail -5 $OUT/tr_$B.log; exit 1; }
  f=$(find $OUT/tr_$B -name 't_kernel_trace.csv' | head -1)
  python3 $R/scripts/unet_breakdown.py $f $B > $OUT/unet_B$B.txt 2>&1 || { cat $OUT/unet_B$B.txt; exit 1; }
  rm -rf $OUT/tr_$B
done
