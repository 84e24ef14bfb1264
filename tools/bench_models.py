#!/usr/bin/env python
"""Throughput of the other generators (fp32 NHWC parity path) at 1080p on one GPU, with the CPU oracle beside
them on a bounded sample.  Not the headline metric (bench.py); numbers go to DESIGN.md / profiles/."""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch

import mon_amd
from oracle import generators as og

GFLOP = {"unet32": 634.5, "unet32_ps": None, "unet64": 2530.0, "llunet++": 11495.7, "retinexnet": 1374.5, "zero_dce++": 41.8}
CASES = [
    ("zero_dce++", dict(name="zero_dce++_re", seed=1), 4, (1080, 1920)),
    ("unet32", dict(name="unet", seed=1, base_channels=32, num_classes=3, enhance=True), 8, (1080, 1920)),
    ("unet32_ps", dict(name="unet", seed=1, base_channels=32, num_classes=3, enhance=True, upsampler="pixelshuffle"), 8, (1080, 1920)),
    ("retinexnet", dict(name="retinexnet", seed=1), 2, (1080, 1920)),
    ("llunet++", dict(name="llunet++_re", seed=1), 1, (1088, 1920)),
    ("hinet64", dict(name="hinet_re", seed=1), 1, (1088, 1920)),
    ("unet64", dict(name="unet", seed=1, base_channels=64, num_classes=3, enhance=True), 8, (1080, 1920)),
]
cdt = "f32"
args = sys.argv[1:]
if args and args[0] in ("f32", "f16", "bf16"):
    cdt = args.pop(0)
only = args
for tag, cfg, n, hw in CASES:
    if only and tag not in only:
        continue
    if False:
        continue                      # no 16-bit graph yet
    m = mon_amd.MODELS.build(compute_dtype=cdt, **cfg).to("cuda") if cdt != "f32" else mon_amd.MODELS.build(**cfg).to("cuda")
    x = torch.rand(n, 3, *hw, device="cuda")
    for _ in range(1 if cdt == "f32" else 4):      # first call packs the weights on the host; then let the clocks ramp up
        m.forward({"image": x})
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    reps = 2 if cdt == "f32" else 10
    for _ in range(reps):
        out = m.forward({"image": x})
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / reps
    rec = {"model": tag, "dtype": cdt, "frames": n, "hw": hw, "ms_per_frame": round(dt / n * 1e3, 2), "frames_per_s": round(n / dt, 3)}
    if GFLOP.get(tag):
        rec["tflops"] = round(GFLOP[tag] * n / dt / 1e3, 2)
    print(json.dumps(rec), flush=True)
    del m, x, out
    torch.cuda.empty_cache()
