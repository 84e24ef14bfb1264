"""Error of the 16-bit matrix-core path of the generators vs the CPU fp32 oracle (diagnostic)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import mon_amd
from oracle import generators as og
from tests.helpers import rel_err, psnr

def run(tag, build, ref_fn, x, keys):
    for dt in ("f32", "f16", "bf16"):
        m = mon_amd.MODELS.build(compute_dtype=dt, **build).to("cuda")
        out = m.forward({"image": x.cuda()})
        ref = ref_fn(m.state_dict(), x)
        print(tag, dt, {k: ("%.2e" % rel_err(out[k].cpu(), ref[k]), "%.1f dB" % psnr(out[k].cpu(), ref[k])) for k in keys}, flush=True)

g = torch.Generator().manual_seed(2)
x = torch.rand(2, 3, 135, 70, generator=g)
run("unet32", dict(name="unet", seed=3, base_channels=32, num_classes=3, enhance=True), lambda sd, t: og.unet_forward(sd, t, "bilinear", True), x, ["enhanced"])
x = torch.rand(2, 3, 48, 64, generator=g)
run("unet32ps", dict(name="unet", seed=3, base_channels=32, num_classes=3, enhance=True, upsampler="pixelshuffle"), lambda sd, t: og.unet_forward(sd, t, "pixelshuffle", True), x, ["enhanced"])
x = torch.rand(2, 3, 45, 67, generator=g)
run("retinex", dict(name="retinexnet", seed=3), lambda sd, t: og.retinexnet_forward(sd, t), x, ["reflectance", "illumination", "illumination_delta", "enhanced"])
x = torch.rand(1, 3, 32, 48, generator=g)
run("llunet++", dict(name="llunet++_re", seed=31), lambda sd, t: og.llunetpp_forward(sd, t), x, ["enhanced"])
x = torch.rand(2, 3, 32, 48, generator=g)
run("hinet16", dict(name="hinet_re", seed=5, num_channels=16), lambda sd, t: og.hinet_forward(sd, t), x, ["stage1", "enhanced"])
run("zero_ig", dict(name="zero_ig_enhance", seed=5), lambda sd, t: og.zero_ig_enhance_forward(sd, t), x, ["enhanced"])
x = torch.rand(2, 3, 40, 56, generator=g)
run("zero_dce++", dict(name="zero_dce++_re", seed=5), lambda sd, t: og.zero_dcepp_forward(sd, t), x, ["adjust", "enhanced"])
run("zero_dce++s2", dict(name="zero_dce++_re", seed=5, scale_factor=2.0), lambda sd, t: og.zero_dcepp_forward(sd, t, 8, 2.0), x, ["adjust", "enhanced"])
