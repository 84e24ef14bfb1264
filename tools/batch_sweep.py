#!/usr/bin/env python
"""Frames/s of one model at 1080p as a function of frames per call, through a captured HIP graph (no Python launch time):
does a smaller working set (activations resident in the 256 MB Infinity Cache) beat fuller grids?"""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import mon_amd

name = sys.argv[1] if len(sys.argv) > 1 else "unet32"
CFG = {"unet32": dict(name="unet", base_channels=32, num_classes=3, enhance=True),
       "unet32_ps": dict(name="unet", base_channels=32, num_classes=3, enhance=True, upsampler="pixelshuffle"),
       "unet64": dict(name="unet", base_channels=64, num_classes=3, enhance=True),
       "retinexnet": dict(name="retinexnet")}[name]
m = mon_amd.MODELS.build(seed=1, compute_dtype="bf16", **CFG).to("cuda")
for b in (1, 2, 3, 4, 8, 16):
    x = torch.rand(b, 3, 1080, 1920, device="cuda")
    run = m.capture_graph(x)
    for _ in range(3):
        run(x)
    torch.cuda.synchronize()
    reps = max(4, 48 // b)
    t0 = time.perf_counter()
    for _ in range(reps):
        run(x)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / reps
    print(json.dumps({"model": name, "frames_per_call": b, "ms_per_call": round(dt * 1e3, 3), "frames_per_s": round(b / dt, 1)}), flush=True)
    del run, x
    torch.cuda.empty_cache()
