#!/usr/bin/env python
"""The nine layers of U-Net-32 (batch 32, 1080p) that run on the register-weights 3x3 kernel (mconv3_kernel), one by one: microseconds and TFLOP/s.
A/B switches are the library's environment variables (IE_MCONV_TH, IE_MCONV_WGS, IE_MCONV_POOL ...): python tools/bench_unet32_deep.py [f16|bf16]"""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from mon_amd import ops
CL = torch.channels_last
dt = torch.float16 if (len(sys.argv) < 2 or sys.argv[1] == "f16") else torch.bfloat16
N = 32
LAYERS = [("down2.c1", 270, 480, [64], 128, False), ("down2.c2+pool", 270, 480, [128], 128, True), ("down3.c1", 135, 240, [128], 256, False),
          ("down3.c2+pool", 135, 240, [256], 256, True), ("down4.c1", 67, 120, [256], 256, False), ("up1.c1", 135, 240, [256, 256], 256, False),
          ("up1.c2", 135, 240, [256], 128, False), ("up2.c1", 270, 480, [128, 128], 128, False)]
tot = 0.0
for name, h, w, cs, co, pool in LAYERS:
    xs = [torch.randn(N, c, h, w, device="cuda", dtype=dt).contiguous(memory_format=CL) for c in cs]
    mw = ops.MConvWeights(torch.randn(co, sum(cs), 3, 3) / (sum(cs) * 9) ** 0.5, torch.zeros(co), cs, dt, "cuda")
    run = lambda: ops.mconv2d(xs, mw, act="relu", pool=pool)
    for _ in range(3): run()
    ts = []
    for _ in range(7):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); run(); e1.record(); torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) * 1e3)
    us = sorted(ts)[len(ts) // 2]
    tot += us
    print(json.dumps({"layer": name, "us": round(us), "tflops": round(2.0 * N * h * w * sum(cs) * co * 9 / us / 1e6)}), flush=True)
    del xs, mw
print(json.dumps({"sum_us": round(tot), "env": {k: v for k, v in os.environ.items() if k.startswith("IE_MCONV")}}), flush=True)
