#!/bin/bash
# Per-layer kernel durations with the tile height forced (CLD_TILING_HALF = 0 / 1 / 2) at one batch size:
#   bash scripts/tile_height_ab.sh 1024      -> gpurun_out/tile_ab/h{0,1,2}_B<agents>.csv
# (input to the cost model of pick_tiling in csrc/cld_api.hip)
# The tiling overrides are compiled only into -DCLD_EXPERIMENTS builds (the shipped library reads no environment variable):
#   CLD_LIB_OUT=libcld_hip_exp.so CLD_EXTRA_FLAGS=-DCLD_EXPERIMENTS bash controllable-latent-diffusion-for-traffic-simulation_amd/build.sh
set -u
B=${1:-1024}
R=$GRAFT_REPO_ROOT
export CLD_LIB_PATH=$R/controllable-latent-diffusion-for-traffic-simulation_amd/libcld_hip_exp.so
[ -f "$CLD_LIB_PATH" ] || { echo "build the -DCLD_EXPERIMENTS library first (see the header of this script)"; exit 1; }
OUT=$R/gpurun_out/tile_ab
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for h in 0 1 2; do
  export CLD_TILING_HALF=$h CLD_TILING_C=0
  timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/h${h}_$B -o t -- python3 $R/scripts/sweep_batch.py $B > $OUT/h${h}_$B.log 2>&1 || exit 1
  cp $OUT/h${h}_$B/t_kernel_stats.csv $OUT/h${h}_B$B.csv
done
