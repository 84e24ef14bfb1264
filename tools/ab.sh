#!/bin/bash
# A/B builds of the library on the same box: tools/ab.sh lib1.so lib2.so ...   (N=16 frames bf16)
for lib in "$@"; do
  echo "== $lib"; IE_HIP_LIB=$(readlink -f $lib) python tools/tune1.py ${N:-16} bf16 2>&1 | grep -v amdgpu.ids
done
