#!/usr/bin/env python
"""Per-layer throughput of the 16-bit weight gradient (ie_convn_wgrad_nhwc, matrix-core kernel) at U-Net-64 shapes (diagnostic).
IE_WGRAD_FMA=1 times the fp32-FMA kernel on the same tensors."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch

from mon_amd import ops

CL = torch.channels_last
dt = torch.bfloat16 if (len(sys.argv) < 2 or sys.argv[1] == "bf16") else torch.float16
SHAPES = [  # (name, N, H, W, [cin...], cout)
    ("16->64 full (RGB edge)", 2, 1080, 1920, [16], 64),
    ("64->64 full", 2, 1080, 1920, [64], 64),
    ("64+64->64 full", 2, 1080, 1920, [64, 64], 64),
    ("128->128 half", 2, 540, 960, [128], 128),
    ("256->256 quarter", 2, 270, 480, [256], 256),
    ("512->512 eighth", 2, 135, 240, [512], 512),
    ("512+512->256 eighth", 2, 135, 240, [512, 512], 256),
]
only = sys.argv[2:]
for name, n, h, w, cs, co in SHAPES:
    if only and not any(o in name for o in only):
        continue
    xs = [torch.randn(n, c, h, w, device="cuda", dtype=dt).contiguous(memory_format=CL) for c in cs]
    dy = torch.randn(n, co, h, w, device="cuda", dtype=dt).contiguous(memory_format=CL)
    cw = ops.ConvWeightsView(None, None, co, sum(cs), 3, 3)
    dw = torch.empty(9 * sum(cs) * co, device="cuda")
    for _ in range(2):
        ops.conv2d_wgrad(xs, dy, cw, dw, None, 1, 1)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    reps = 5
    e0.record()
    for _ in range(reps):
        ops.conv2d_wgrad(xs, dy, cw, dw, None, 1, 1)
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / reps
    flop = 2.0 * n * h * w * sum(cs) * co * 9
    byts = n * h * w * (sum(cs) + co) * 2
    print(json.dumps({"layer": name, "ms": round(ms, 3), "tflops": round(flop / ms / 1e9, 1), "alg_GBps": round(byts / ms / 1e6, 1)}), flush=True)
    del xs, dy, dw
