#!/bin/bash
# Build libcld_hip.so for gfx950 (MI355X) in-tree.  hipcc cross-compiles without a GPU.
set -euo pipefail
cd "$(dirname "$0")"
HIPCC=${HIPCC:-/opt/rocm/bin/hipcc}
"$HIPCC" --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared \
    -o "${CLD_LIB_OUT:-libcld_hip.so}" csrc/conv_block.hip csrc/misc_kernels.hip csrc/context_kernels.hip csrc/guide_kernels.hip csrc/cld_api.hip ${CLD_EXTRA_FLAGS:-} "$@"
echo "built $(pwd)/${CLD_LIB_OUT:-libcld_hip.so}"
