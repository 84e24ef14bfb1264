import os, sys, time, json
sys.path.insert(0, "/root/repo")
import torch, mon_amd
n = int(sys.argv[1]); dt = sys.argv[2]
m = mon_amd.MODELS.build(name="unet", seed=1, base_channels=32, num_classes=3, enhance=True, compute_dtype=dt).to("cuda")
x = torch.rand(n, 3, 1080, 1920, device="cuda")
for _ in range(3): m.forward({"image": x})
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(6): out = m.forward({"image": x})
torch.cuda.synchronize(); d = (time.perf_counter() - t0) / 6
print(json.dumps({"frames": n, "dt": dt, "fps": round(n / d, 1), "env": {k: v for k, v in os.environ.items() if k.startswith("IE_UNET")}}), flush=True)
