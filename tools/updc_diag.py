#!/usr/bin/env python
"""Diagnostics for ie_unet_up_head_nhwc: identity head / identity second layer isolate the stages; prints where the result differs."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from mon_amd import ops

CL = torch.channels_last
tdt = torch.float16
n, h, w = 1, 12, 150
g = torch.Generator().manual_seed(41)
skip = torch.randn(n, 32, h, w, generator=g).to(tdt).cuda().contiguous(memory_format=CL)
up = torch.randn(n, 32, h, w, generator=g).to(tdt).cuda().contiguous(memory_format=CL)
for mode in ("idB", "full"):
    wa = (torch.randn(32, 64, 3, 3, generator=g) / 24.0).to(tdt).float()
    wb = (torch.randn(32, 32, 3, 3, generator=g) / 17.0).to(tdt).float()
    if mode == "idB":
        wb = torch.zeros(32, 32, 3, 3); wb[torch.arange(32), torch.arange(32), 1, 1] = 1.0
    mwa = ops.MConvWeights(wa, torch.zeros(32), [32, 32], tdt, "cuda")
    mwb = ops.MConvWeights(wb, torch.zeros(32), [32], tdt, "cuda")
    for half in (0, 1):
        w1 = torch.zeros(16, 32, 1, 1); w1[torch.arange(16), torch.arange(16) + 16 * half, 0, 0] = 1.0
        tail = ops.MTailWeights(w1, torch.zeros(16), tdt, "cuda")
        got = ops.unet_up_head(skip, up, mwa, mwb, tail)
        mid = ops.mconv2d([skip, up], mwa, act="relu")
        x2 = ops.mconv2d(mid, mwb, act="relu")
        want = x2.float()[:, 16 * half:16 * half + 16].contiguous()
        d = (got - want).abs()
        bad = d > 1e-3
        print(mode, "channels", 16 * half, "max err", float(d.max()), "bad frac", float(bad.float().mean()))
        if bad.any():
            print("  bad per channel:", bad.sum((0, 2, 3)).tolist())
            print("  bad per row    :", bad.sum((0, 1, 3)).tolist())
            cols = bad.sum((0, 1, 2)).tolist()
            print("  bad per column :", cols)
            print("  got[0,0,3,:12] ", got[0, 0, 3, :12].tolist())
            print("  want[0,0,3,:12]", want[0, 0, 3, :12].tolist())
