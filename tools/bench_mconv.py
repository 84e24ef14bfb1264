#!/usr/bin/env python
"""Per-layer throughput of ie_mconv_nhwc at the shapes of U-Net-32/64 and RetinexNet at 1080p (diagnostic)."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch

from mon_amd import ops

CL = torch.channels_last
dt = torch.bfloat16 if (len(sys.argv) < 2 or sys.argv[1] == "bf16") else torch.float16
SHAPES = [  # (name, N, H, W, [cin...], cout, k)
    ("32->32 full", 4, 1080, 1920, [32], 32, 3),
    ("64->64 full", 4, 1080, 1920, [64], 64, 3),
    ("64+64->64 full", 2, 1080, 1920, [64, 64], 64, 3),
    ("128->128 half", 4, 540, 960, [128], 128, 3),
    ("256->256 quarter", 4, 270, 480, [256], 256, 3),
    ("512->512 eighth", 4, 135, 240, [512], 512, 3),
    ("8->32 first", 4, 1080, 1920, [8], 32, 3),
    ("64->3 1x1", 4, 1080, 1920, [64], 3, 1),
    ("8->64 9x9", 2, 1080, 1920, [8], 64, 9),
]
only = sys.argv[2:]
for name, n, h, w, cs, co, k in SHAPES:
    if only and not any(o in name for o in only):
        continue
    g = torch.Generator().manual_seed(0)
    xs = [torch.randn(n, c, h, w, device="cuda", dtype=dt).contiguous(memory_format=CL) for c in cs]
    wt = torch.randn(co, sum(cs), k, k) / (sum(cs) * k * k) ** 0.5
    mw = ops.MConvWeights(wt, torch.zeros(co), cs, dt, "cuda")
    for _ in range(2):
        y = ops.mconv2d(xs, mw, pad=k // 2, act="relu")
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    reps = 5
    e0.record()
    for _ in range(reps):
        y = ops.mconv2d(xs, mw, pad=k // 2, act="relu")
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / reps
    flop = 2.0 * n * h * w * sum(cs) * co * k * k
    byts = n * h * w * (sum(cs) + co) * 2
    print(json.dumps({"layer": name, "ms": round(ms, 3), "tflops": round(flop / ms / 1e9, 1), "alg_GBps": round(byts / ms / 1e6, 1)}), flush=True)
    del xs, y, mw
