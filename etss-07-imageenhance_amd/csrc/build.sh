#!/bin/bash
# Build libie_hip.so (gfx950) in-tree: one hipcc per source in parallel, then link.
# Usage: build.sh [extra hipcc flags];  IE_OUT=path overrides the output, IE_JOBS the parallelism.
set -e
cd "$(dirname "$0")"
OUT=${IE_OUT:-../libie_hip.so}
SRCS="ie_core.hip ie_conv_direct.hip ie_ops.hip ie_conv_mfma.hip ie_ops16.hip ie_norm.hip ie_train.hip ie_wgrad_mfma.hip ie_zref.hip ie_zdce_kernels.hip ie_zdce_fused.hip ie_zdce_plan.hip"
OBJ=$(mktemp -d)
trap 'rm -rf "$OBJ"' EXIT
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -fvisibility=hidden -Wall -Wno-unused-function $*"
printf '%s\n' $SRCS | xargs -P "${IE_JOBS:-8}" -I{} sh -c "hipcc $FLAGS -c {} -o $OBJ/{}.o"
hipcc --offload-arch=gfx950 -shared -fPIC -o "$OUT" $OBJ/*.o
echo "built $(readlink -f $OUT)"
