#!/bin/bash
# A/B builds of ONE kernel file: tools/exp_build.sh <source.hip> <tag> [extra hipcc flags] -> etss-07-imageenhance_amd/libie_<tag>.so
# (the other objects come from the default build's cache; run csrc/build.sh first).  Select at run time with IE_HIP_LIB=<path>.
set -e
cd "$(dirname "$0")/../etss-07-imageenhance_amd/csrc"
SRC=$1; TAG=$2; shift 2
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -fvisibility=hidden -Wall -Wno-unused-function "
KEY=$(printf '%s' "$FLAGS" | md5sum | cut -c1-10)
OBJ=_obj/$KEY
[ -f "$OBJ/$SRC.o" ] || { echo "default build cache $OBJ missing"; exit 1; }
mkdir -p _obj/exp
hipcc $FLAGS "$@" -I. -c "$SRC" -o "_obj/exp/$TAG.o"
objs=""
for s in $(ls ie_*.hip); do if [ "$s" = "$SRC" ]; then objs="$objs _obj/exp/$TAG.o"; else objs="$objs $OBJ/$s.o"; fi; done
hipcc --offload-arch=gfx950 -shared -fPIC -o "../libie_$TAG.so" $objs
echo "built libie_$TAG.so"
