#!/usr/bin/env python
"""Single-frame serving latency: eager (one launch per kernel from Python) vs a captured HIP graph."""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import mon_amd

for tag, cfg, hw in (("zero_dce bf16", dict(name="zero_dce_re", compute_dtype="bf16"), (1080, 1920)),
                     ("unet32 bf16", dict(name="unet", base_channels=32, num_classes=3, enhance=True, compute_dtype="bf16"), (1080, 1920)),
                     ("unet32 bf16 small", dict(name="unet", base_channels=32, num_classes=3, enhance=True, compute_dtype="bf16"), (256, 256)),
                     ("retinexnet bf16", dict(name="retinexnet", compute_dtype="bf16"), (1080, 1920)),
                     ("zero_dce++ bf16", dict(name="zero_dce++_re", compute_dtype="bf16"), (1080, 1920))):
    m = mon_amd.MODELS.build(seed=1, **cfg).to("cuda")
    x = torch.rand(1, 3, *hw, device="cuda")

    def timed(fn, reps=30):
        for _ in range(3):
            fn()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(reps):
            fn()
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / reps * 1e3
    eager = timed(lambda: m.forward({"image": x}))
    ref = m.forward({"image": x})["enhanced"].clone()
    run = m.capture_graph(x)
    graphed = timed(lambda: run(x))
    same = bool(torch.equal(run(x)["enhanced"], ref))
    print(json.dumps({"model": tag, "hw": hw, "eager_ms": round(eager, 3), "graph_ms": round(graphed, 3), "speedup": round(eager / graphed, 2), "bit_identical": same}), flush=True)
    del m, run
    torch.cuda.empty_cache()
