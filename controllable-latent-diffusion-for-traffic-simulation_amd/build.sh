#!/bin/bash
# Build libcld_hip.so for gfx950 (MI355X) in-tree.  hipcc cross-compiles without a GPU.
# Each .hip file is compiled to its own object (in parallel, only when it or a header changed), then linked.
#   CLD_LIB_OUT      output name (default libcld_hip.so)
#   CLD_EXTRA_FLAGS  extra compile flags, e.g. -DCLD_EXPERIMENTS (environment-variable kernel selection for A/B
#                    experiments) or -DCLD_STAMPS (in-kernel cycle stamps); objects of such builds live in their own directory
set -euo pipefail
cd "$(dirname "$0")"
HIPCC=${HIPCC:-/opt/rocm/bin/hipcc}
OUT=${CLD_LIB_OUT:-libcld_hip.so}
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC ${CLD_EXTRA_FLAGS:-} $*"
TAG=$(echo "$FLAGS" | md5sum | cut -c1-8)
OBJ=build/obj_$TAG
mkdir -p "$OBJ"
SRCS="conv_block conv_chain chain_wino misc_kernels context_kernels wino_kernels wino44_kernels wino1d_kernels wino1d_edge guide_kernels collision_kernels cld_api"
pids=()
for s in $SRCS; do
    src=csrc/$s.hip
    obj=$OBJ/$s.o
    if [ ! -f "$obj" ] || [ "$src" -nt "$obj" ] || [ csrc/cld_kernels.h -nt "$obj" ] || [ csrc/chain_common.h -nt "$obj" ] || [ csrc/wino1d_common.h -nt "$obj" ] || [ ../include/cld.h -nt "$obj" ]; then
        "$HIPCC" $FLAGS -c "$src" -o "$obj" &
        pids+=($!)
    fi
done
for p in "${pids[@]:-}"; do
    [ -n "$p" ] && wait "$p"
done
objs=""
for s in $SRCS; do objs="$objs $OBJ/$s.o"; done
"$HIPCC" --offload-arch=gfx950 -fPIC -shared -o "$OUT" $objs
echo "built $(pwd)/$OUT"
