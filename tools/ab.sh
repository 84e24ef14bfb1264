#!/bin/bash
# A/B two builds of the library on the same box: tools/ab.sh libA.so libB.so [N]
for lib in "$1" "$2"; do
  echo "== $lib"; IE_HIP_LIB=$(readlink -f $lib) IE_ZDCE_RB=${RB:-270} python tools/tune1.py ${3:-16} bf16 2>&1 | grep -v amdgpu.ids
done
