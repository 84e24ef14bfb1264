#!/bin/bash
# time the Zero-DCE kernels with each A/B library: tools/exp_run.sh tag [tag ...]   (on the GPU box)
for t in "$@"; do
  L=$PWD/etss-07-imageenhance_amd/libie_$t.so
  IE_HIP_LIB=$L python bench.py --steps 6 --warmup 2 --no-extras --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
j=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$t', j['value'], j['roofline']['kernel_ms_per_step'])"
done
