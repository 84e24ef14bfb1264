#!/usr/bin/env python
"""Bandwidth of the training graph's BatchNorm passes on their own (U-Net-64 shapes at batch 8, bf16): forward = statistics pass + apply pass, backward = the
recompute form (ie_norm_backward_pre_nhwc: statistics pass over x and dy, apply pass x, dy -> dx).  Prints ms and the HBM GB/s the pass pair moves (algorithmic bytes).
Usage: tools/bench_norm.py [reps]   (environment knobs of csrc/ie_norm.hip apply: IE_NORM_GRID_CAP, IE_NORM_RED_CAP)"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch

import mon_amd  # noqa: F401
from mon_amd import ops

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 10
CL = torch.channels_last
for (n, c, h, w) in ((8, 64, 1080, 1920), (8, 128, 540, 960), (8, 256, 270, 480), (8, 512, 135, 240)):
    x = torch.randn(n, c, h, w, device="cuda", dtype=torch.bfloat16).contiguous(memory_format=CL)
    dy = torch.randn(n, c, h, w, device="cuda", dtype=torch.bfloat16).contiguous(memory_format=CL)
    gamma = torch.rand(c, device="cuda") + 0.5
    beta = torch.randn(c, device="cuda") * 0.1
    dg, db = torch.zeros(c, device="cuda"), torch.zeros(c, device="cuda")
    y, mean, inv = ops.norm_forward(x, gamma, beta, False, act="relu")
    e = [torch.cuda.Event(enable_timing=True) for _ in range(3)]
    for _ in range(2):
        ops.norm_forward(x, gamma, beta, False, act="relu")
        ops.norm_backward(x, dy, y, gamma, mean, inv, False, dg, db, act="relu", beta=beta, recompute=True)
    torch.cuda.synchronize()
    e[0].record()
    for _ in range(reps):
        ops.norm_forward(x, gamma, beta, False, act="relu")
    e[1].record()
    for _ in range(reps):
        ops.norm_backward(x, dy, y, gamma, mean, inv, False, dg, db, act="relu", beta=beta, recompute=True)
    e[2].record()
    torch.cuda.synchronize()
    gb = x.numel() * 2 / 1e9
    f_ms, b_ms = e[0].elapsed_time(e[1]) / reps, e[1].elapsed_time(e[2]) / reps
    print(f"{n}x{c}x{h}x{w} bf16 ({gb:.2f} GB per tensor): forward {f_ms:.3f} ms = {3 * gb / f_ms:.2f} TB/s over 3 tensor passes | backward {b_ms:.3f} ms = {5 * gb / b_ms:.2f} TB/s over 5",
          flush=True)
    del x, dy, y
    torch.cuda.empty_cache()
