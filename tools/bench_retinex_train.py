#!/usr/bin/env python
"""RetinexNet training step (phase decom / relight) at 1080p on the fp32 and the 16-bit graph: frames/s.  Usage: python tools/bench_retinex_train.py [frames]"""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import mon_amd

n = int(sys.argv[1]) if len(sys.argv) > 1 else 2
only = sys.argv[2:]          # e.g. "bf16 relight": one configuration per process (fresh allocator state)
g = torch.Generator(device="cuda").manual_seed(3)
high = torch.rand((n, 3, 1080, 1920), generator=g, device="cuda")
low = (high ** 2.2 * 0.3).contiguous()
for dt in ("f32", "bf16", "f16"):
    for phase in ("decom", "relight"):
        if only and (dt not in only or phase not in only):
            continue
        model = mon_amd.MODELS.build(name="retinexnet", seed=1).to("cuda")
        model.train_phase = phase
        tr = mon_amd.Trainer(model, optimizer={"name": "adam", "lr": 1e-4}, gradient_clip_val=None, compute_dtype=dt)
        for _ in range(3):
            tr.training_step({"image": low, "ref_image": high})
        ts = []          # median of individually synchronised steps: an allocator hiccup (hundreds of ms, once) must not pass for the step time
        for _ in range(9):
            torch.cuda.synchronize(); t0 = time.perf_counter()
            out = tr.training_step({"image": low, "ref_image": high})
            torch.cuda.synchronize(); ts.append(time.perf_counter() - t0)
        d = sorted(ts)[len(ts) // 2]
        print(json.dumps({"model": "retinexnet", "phase": phase, "graph": dt, "frames": n, "ms_per_step": round(d * 1e3, 1), "frames_per_s": round(n / d, 2),
                          "loss": round(float(out["loss"]), 5), "ms_steps": [round(t * 1e3, 1) for t in ts]}), flush=True)
        del tr, model
        torch.cuda.empty_cache()
