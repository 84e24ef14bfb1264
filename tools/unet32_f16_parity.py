import sys; sys.path.insert(0, "/root/repo")
import torch, mon_amd
from oracle import generators as og
torch.manual_seed(0)
g = torch.Generator().manual_seed(11)
x = (torch.rand(1, 3, 1080, 1920, generator=g) ** 2.2 * 0.5)
m = mon_amd.MODELS.build(name="unet", seed=3, compute_dtype="f16", base_channels=32, num_classes=3, enhance=True).to("cuda")
m32 = mon_amd.MODELS.build(name="unet", weights=m.state_dict(), base_channels=32, num_classes=3, enhance=True).to("cuda")
a = m.forward({"image": x.cuda()})["enhanced"].cpu().double()
r = m32.forward({"image": x.cuda()})["enhanced"].cpu().double()
print("unet32 f16 vs fp32 path @1080p: max rel err %.3e" % float((a - r).abs().max() / r.abs().max()))
