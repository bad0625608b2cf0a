#!/bin/bash
# Per-launch U-Net breakdown at one batch size under forced tilings (needs the -DCLD_EXPERIMENTS build):
#   bash scripts/tiling_exp.sh <agents>   -> gpurun_out/tiling_exp/<variant>_B<agents>.txt
set -u
B=${1:-4096}
R=$GRAFT_REPO_ROOT
export CLD_LIB_PATH=$R/controllable-latent-diffusion-for-traffic-simulation_amd/libcld_hip_exp.so
OUT=$R/gpurun_out/tiling_exp; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
run() {  # name, env...
  name=$1; shift
  rm -rf $OUT/tr
  env "$@" true
  ( export "$@"; timeout -k 10 200 rocprofv3 --kernel-trace --output-format csv -d $OUT/tr -o t -- python3 $R/scripts/one_unet.py $B > $OUT/$name.log 2>&1 ) || { tail -3 $OUT/$name.log; return 1; }
  f=$(find $OUT/tr -name 't_kernel_trace.csv' | head -1)
  python3 $R/scripts/unet_breakdown.py $f $B > $OUT/${name}_B$B.txt 2>&1
  tail -1 $OUT/${name}_B$B.txt
  rm -rf $OUT/tr
}
run default CLD_X=0 && run b_h0 CLD_TILING=b CLD_TILING_HALF=0 CLD_TILING_C=0 && run b_h1 CLD_TILING=b CLD_TILING_HALF=1 CLD_TILING_C=0 && run b_h2 CLD_TILING=b CLD_TILING_HALF=2 CLD_TILING_C=0 && run c_all CLD_TILING=b CLD_TILING_C=all
