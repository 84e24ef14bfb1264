#!/bin/bash
# Build libie_hip.so (gfx950) in-tree.  Usage: build.sh [extra hipcc flags]
set -e
cd "$(dirname "$0")"
OUT=${IE_OUT:-../libie_hip.so}
SRCS="ie_core.hip ie_conv_direct.hip ie_ops.hip ie_conv_mfma.hip ie_ops16.hip ie_norm.hip ie_train.hip ie_zdce_kernels.hip ie_zdce_fused.hip ie_zdce_plan.hip"
hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -fvisibility=hidden \
      -Wall -Wno-unused-function -o "$OUT" $SRCS "$@"
echo "built $(readlink -f $OUT)"
