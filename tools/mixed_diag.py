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
tr = mon_amd.Trainer(model, optimizer={"lr": 1e-3, "eps": 1e-3}, loss={"name": "l1_ssim_loss"}, compute_dtype=dt)
out = tr.training_step({"image": img, "ref_image": tgt})
print("loss", float(out["loss"]), log[0]["loss"][0])
gr = tr.gradients()
for k, ref in log[0]["grads"].items():
    a, b = gr[k].double().reshape(-1), ref.reshape(-1)
    print(f"{k:28s} cos {float((a*b).sum()/(a.norm()*b.norm()+1e-30)):7.4f}  norm ratio {float(a.norm()/(b.norm()+1e-30)):6.3f}")
