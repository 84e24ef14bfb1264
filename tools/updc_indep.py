#!/usr/bin/env python
"""Batch independence / determinism of the fused last U-Net level at 1080p: where do a frame alone and the same frame inside a batch differ?"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import mon_amd

dt = sys.argv[1] if len(sys.argv) > 1 else "bf16"
g = torch.Generator().manual_seed(11)
x = (torch.rand(2, 3, 1080, 1920, generator=g) ** 2.2 * 0.5).cuda()
m = mon_amd.MODELS.build(name="unet", seed=3, compute_dtype=dt, base_channels=32, num_classes=3, enhance=True).to("cuda")
a = m.forward({"image": x})["enhanced"]
for rep in range(3):
    b = m.forward({"image": x})["enhanced"]
    print("repeat", rep, "equal", bool(torch.equal(a, b)))
one = m.forward({"image": x[1:2]})["enhanced"]
d = (a[1:2] - one).abs()
print("batch vs single: max", float(d.max()), "count", int((d > 0).sum()))
if (d > 0).any():
    rows = (d > 0).sum((0, 1, 3)).nonzero().flatten().tolist()
    cols = (d > 0).sum((0, 1, 2)).nonzero().flatten().tolist()
    print("rows", rows[:40], "... n =", len(rows))
    print("cols", cols[:40], "... n =", len(cols))
