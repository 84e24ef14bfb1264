"""Why do some RetinexNet training steps take 0.4 - 1.4 s?  Per step: wall time, torch allocator segment allocations / frees, reserved bytes, gc collections.
python tools/retinex_train_diag.py [decom|relight] [steps]"""
import gc, json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import mon_amd
phase = sys.argv[1] if len(sys.argv) > 1 else "relight"
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 16
g = torch.Generator(device="cuda").manual_seed(3)
high = torch.rand((2, 3, 1080, 1920), generator=g, device="cuda")
low = (high ** 2.2 * 0.3).contiguous()
model = mon_amd.MODELS.build(name="retinexnet", seed=1).to("cuda")
model.train_phase = phase
tr = mon_amd.Trainer(model, optimizer={"name": "adam", "lr": 1e-4}, gradient_clip_val=None, compute_dtype="bf16")
for i in range(steps):
    s0 = torch.cuda.memory_stats()
    g0 = [c["collections"] for c in gc.get_stats()]
    torch.cuda.synchronize(); t0 = time.perf_counter()
    tr.training_step({"image": low, "ref_image": high})
    t1 = time.perf_counter()
    torch.cuda.synchronize(); t2 = time.perf_counter()
    s1 = torch.cuda.memory_stats()
    g1 = [c["collections"] for c in gc.get_stats()]
    print(json.dumps({"step": i, "ms": round((t2 - t0) * 1e3, 1), "host_ms": round((t1 - t0) * 1e3, 1),
                      "seg_alloc": s1["num_device_alloc"] - s0["num_device_alloc"], "seg_free": s1["num_device_free"] - s0["num_device_free"],
                      "retries": s1["num_alloc_retries"] - s0["num_alloc_retries"], "reserved_GiB": round(s1["reserved_bytes.all.current"] / 2 ** 30, 2),
                      "gc": [b - a for a, b in zip(g0, g1)]}), flush=True)
