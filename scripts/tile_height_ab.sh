#!/bin/bash
# Per-layer kernel durations with the tile height forced (CLD_TILING_HALF = 0 / 1 / 2) at one batch size:
#   bash scripts/tile_height_ab.sh 1024      -> gpurun_out/tile_ab/h{0,1,2}_B<agents>.csv
# (input to the cost model of pick_tiling in csrc/cld_api.hip)
set -u
B=${1:-1024}
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/tile_ab
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for h in 0 1 2; do
  export CLD_TILING_HALF=$h CLD_TILING_C=0
  timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/h${h}_$B -o t -- python3 $R/scripts/sweep_batch.py $B > $OUT/h${h}_$B.log 2>&1 || exit 1
  cp $OUT/h${h}_$B/t_kernel_stats.csv $OUT/h${h}_B$B.csv
done
