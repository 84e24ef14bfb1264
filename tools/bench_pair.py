#!/usr/bin/env python
"""ie_mconv_pair_nhwc against two ie_mconv_nhwc launches (64 -> 64 -> 64, 3x3, ReLU) at 1080p: ms and TFLOP/s.  python tools/bench_pair.py [frames] [f16|bf16] [zeros|replicate]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from mon_amd import ops
CL = torch.channels_last
n = int(sys.argv[1]) if len(sys.argv) > 1 else 8
tdt = ops.TORCH_DT[sys.argv[2] if len(sys.argv) > 2 else "f16"]
pm = sys.argv[3] if len(sys.argv) > 3 else "replicate"
g = torch.Generator().manual_seed(1)
x = torch.randn(n, 64, 1080, 1920, generator=g).to(tdt).cuda().contiguous(memory_format=CL)
mws = [ops.MConvWeights((torch.randn(64, 64, 3, 3, generator=g) / 24.0), torch.randn(64, generator=g) * 0.2, [64], tdt, "cuda") for _ in range(2)]
def timed(fn, reps=10):
    for _ in range(3): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(reps): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / reps
fl = 2 * 2.0 * n * 1080 * 1920 * 64 * 64 * 9
t2 = timed(lambda: ops.mconv2d(ops.mconv2d(x, mws[0], pad_mode=pm, act="relu"), mws[1], pad_mode=pm, act="relu"))
t1 = timed(lambda: ops.mconv2d_pair(x, mws[0], mws[1], pad_mode=pm))
print(f"{n} frames {pm}: two launches {t2*1e3:.3f} ms = {fl/t2/1e12:.0f} TFLOP/s | pair {t1*1e3:.3f} ms = {fl/t1/1e12:.0f} TFLOP/s")
