"""Per-parameter gradient error of one U-Net training step: HIP path vs torch CPU fp32 and fp64 (diagnostic)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import mon_amd
from oracle import train as ot

ups = sys.argv[1] if len(sys.argv) > 1 else "bilinear"
h, w = 48, 64
model = mon_amd.MODELS.build(name="unet", in_channels=3, num_classes=3, base_channels=8, upsampler=ups, enhance=True, seed=11).to("cuda")
sd0 = model.state_dict()
g = torch.Generator().manual_seed(3)
tgt = torch.rand(2, 3, h, w, generator=g)
img = (tgt ** 2.2 * 0.4).contiguous()
f = lambda p, b, x: ot.unet_forward_train(p, b, x, ups, True)
log32, _ = ot.training_steps(sd0, [(img, tgt)], f, ot.l1_ssim_loss, lr=1e-3)
sd64 = {k: v.double() for k, v in sd0.items()}
log64, _ = ot.training_steps(sd64, [(img.double(), tgt.double())], f, ot.l1_ssim_loss, lr=1e-3)
tr = mon_amd.Trainer(model, optimizer={"lr": 1e-3}, loss={"name": "l1_ssim_loss"})
out = tr.training_step({"image": img, "ref_image": tgt})
gr = tr.gradients()
print("loss hip %.7f  cpu32 %.7f  cpu64 %.7f" % (float(out["loss"]), log32[0]["loss"][0], log64[0]["loss"][0]))
for k in gr:
    r64 = log64[0]["grads"][k]
    m = float(r64.abs().max())
    e_hip = float((gr[k].double() - r64).abs().max()) / m
    e_cpu = float((log32[0]["grads"][k].double() - r64).abs().max()) / m
    print(f"{k:28s} max|g| {m:9.3e}  hip-vs-64 {e_hip:9.2e}  cpu32-vs-64 {e_cpu:9.2e}")
