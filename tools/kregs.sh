#!/bin/bash
# Compile one csrc/*.hip for gfx950 into /tmp and print VGPR / scratch / spill counts of every kernel in it.
# Usage: tools/kregs.sh ie_zdce_x.hip [extra hipcc flags]
set -e
cd "$(dirname "$0")/../etss-07-imageenhance_amd/csrc"
f=$1; shift
b=$(basename "$f" .hip)
hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -fvisibility=hidden -Wall -Wno-unused-function -I. "$@" -c "$f" -o /tmp/$b.o -save-temps=obj
grep -E "^\s+\.(vgpr_count|private_segment_fixed_size|vgpr_spill_count|name):" /tmp/$b-hip-amdgcn-amd-amdhsa-gfx950.s | paste - - - - | sed -E 's/\s+/ /g'
