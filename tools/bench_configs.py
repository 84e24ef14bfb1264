#!/usr/bin/env python
"""BASELINE.json configs[2] and configs[3] at their stated batch sizes on ONE GPU (frames/s, peak memory)."""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import mon_amd

for tag, cfg, n in (("C3 unet32+pixelshuffle batch=32 1080p bf16", dict(name="unet", base_channels=32, num_classes=3, enhance=True, upsampler="pixelshuffle"), 32),
                    ("C3' unet32 bilinear batch=32 1080p bf16", dict(name="unet", base_channels=32, num_classes=3, enhance=True), 32),
                    ("C4 retinexnet batch=32/GPU 1080p bf16", dict(name="retinexnet"), 32)):
    torch.cuda.reset_peak_memory_stats()
    m = mon_amd.MODELS.build(seed=1, compute_dtype="bf16", **cfg).to("cuda")
    x = torch.rand(n, 3, 1080, 1920, device="cuda")
    for _ in range(2):
        out = m.forward({"image": x})
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    reps = 3
    for _ in range(reps):
        out = m.forward({"image": x})
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / reps
    print(json.dumps({"config": tag, "frames_per_s": round(n / dt, 1), "ms_per_batch": round(dt * 1e3, 1), "peak_mem_GiB": round(torch.cuda.max_memory_allocated() / 2 ** 30, 1),
                      "finite": bool(torch.isfinite(out["enhanced"]).all())}), flush=True)
    del m, x, out
    torch.cuda.empty_cache()
