"""Per-parameter gradient agreement of the 16-bit training graph vs torch float64 (diagnostic)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, mon_amd
from oracle import train as ot
dt = sys.argv[1] if len(sys.argv) > 1 else "bf16"
model = mon_amd.MODELS.build(name="unet", in_channels=3, num_classes=3, base_channels=16, enhance=True, seed=13).to("cuda")
sd0 = model.state_dict()
g = torch.Generator().manual_seed(14)
tgt = torch.rand(2, 3, 48, 64, generator=g); img = (tgt ** 2.2 * 0.4).contiguous()
log, _ = ot.training_steps({k: v.double() for k, v in sd0.items()}, [(img.double(), tgt.double())], lambda p, b, x: ot.unet_forward_train(p, b, x, "bilinear", True), ot.l1_ssim_loss, lr=1e-3, eps=1e-3)
tdt = {"bf16": torch.bfloat16, "f16": torch.float16}[dt]
logq, _ = ot.training_steps({k: v.double() for k, v in sd0.items()}, [(img.double(), tgt.double())], lambda p, b, x: ot.unet_forward_train(p, b, x, "bilinear", True, q=ot.storage_rounding(tdt)), ot.l1_ssim_loss, lr=1e-3, eps=1e-3)
tr = mon_amd.Trainer(model, optimizer={"lr": 1e-3, "eps": 1e-3}, loss={"name": "l1_ssim_loss"}, compute_dtype=dt)
out = tr.training_step({"image": img, "ref_image": tgt})
print("loss", float(out["loss"]), "float64", log[0]["loss"][0], "float64 with", dt, "storage rounding", logq[0]["loss"][0])
gr = tr.gradients()
cs = lambda a, b: float((a * b).sum() / (a.norm() * b.norm() + 1e-30))
for k, ref in log[0]["grads"].items():
    a, b, c = gr[k].double().reshape(-1), ref.reshape(-1), logq[0]["grads"][k].reshape(-1)
    print(f"{k:28s} engine vs float64: cos {cs(a, b):7.4f} norm ratio {float(a.norm()/(b.norm()+1e-30)):6.3f} | engine vs rounded oracle: cos {cs(a, c):7.4f} | rounded oracle vs float64: cos {cs(c, b):7.4f}")
fa = torch.cat([gr[k].double().reshape(-1) for k in log[0]["grads"]]); fb = torch.cat([v.reshape(-1) for v in log[0]["grads"].values()]); fc = torch.cat([v.reshape(-1) for v in logq[0]["grads"].values()])
print(f"WHOLE GRADIENT  engine vs float64 {cs(fa, fb):.4f} | engine vs rounded oracle {cs(fa, fc):.4f} | rounded oracle vs float64 {cs(fc, fb):.4f}")
