#!/bin/bash
R=$GRAFT_REPO_ROOT
export CLD_LIB_PATH=$R/controllable-latent-diffusion-for-traffic-simulation_amd/libcld_hip_exp.so
OUT=$R/gpurun_out/r04m
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
run() {  # tag, env...
  tag=$1; shift
  env "$@" true
  ( export "$@"; timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/$tag -o t -- python3 $R/scripts/one_unet.py 4096 > $OUT/$tag.log 2>&1 )
  f=$(find $OUT/$tag -name 't_kernel_stats.csv' | head -1)
  echo "== $tag"; grep -E "conv_block_kernel|conv_pair_kernel" $f | awk -F, '{printf "%-110s calls %s avg %.1f us\n", substr($1,1,110), $2, $4/1000}'
}
run base CLD_TILING_C=0
run b_h0 CLD_TILING=b CLD_TILING_HALF=0 CLD_TILING_C=0
run b_h1 CLD_TILING=b CLD_TILING_HALF=1 CLD_TILING_C=0
run b_h2 CLD_TILING=b CLD_TILING_HALF=2 CLD_TILING_C=0
