#!/bin/bash
# A/B of library builds on the ContextEncoder's Winograd kernels: per-kernel average durations (rocprofv3 --kernel-trace --stats) of
# scripts/ctx_time.py 256 for each libcld_<name>.so given.   usage (on the GPU box): bash scripts/wino_ab.sh hip ring8 ...
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/wino_ab
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for n in "$@"; do
    export CLD_LIB_PATH=$R/controllable-latent-diffusion-for-traffic-simulation_amd/libcld_$n.so
    timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/$n -o k -- python3 $R/scripts/ctx_time.py ${B:-256} winograd > $OUT/$n.log 2>&1 || exit 1
    echo "== $n: $(grep structured $OUT/$n.log)"
    grep wino_conv $OUT/$n/k_kernel_stats.csv | awk -F, '{printf "   %s calls %s avg %.1f us\n", $1, $2, $4/1000}'
done
