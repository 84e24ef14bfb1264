import os, sys, time, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, mon_amd
n = int(sys.argv[1]); dt = sys.argv[2]
m = mon_amd.MODELS.build(name="retinexnet", seed=1, compute_dtype=dt).to("cuda")
x = torch.rand(n, 3, 1080, 1920, device="cuda")
for _ in range(2): m.forward({"image": x})
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(4): out = m.forward({"image": x})
torch.cuda.synchronize(); d = (time.perf_counter() - t0) / 4
print(json.dumps({"model": "retinexnet", "frames": n, "dt": dt, "fps": round(n / d, 1)}), flush=True)
