import sys, time; sys.path.insert(0, "/root/repo")
import torch, mon_amd
m = mon_amd.MODELS.build(name="zero_dce_re", seed=1, compute_dtype="bf16").to("cuda")
x = torch.rand(16, 3, 1080, 1920, device="cuda")
for want in (dict(), dict(want_adjust=True), dict(want_enhanced=False, want_u8=True)):
    for _ in range(3): m.enhance(x, **want)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(10): m.enhance(x, **want)
    torch.cuda.synchronize(); d = (time.perf_counter() - t0) / 10
    print(want, "%.3f ms per 16 frames = %.0f frames/s" % (d * 1e3, 16 / d), flush=True)
u8 = (x.permute(0, 2, 3, 1) * 255).to(torch.uint8).contiguous()
for _ in range(3): m.enhance_u8(u8)
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(10): m.enhance_u8(u8)
torch.cuda.synchronize(); d = (time.perf_counter() - t0) / 10
print("enhance_u8 (uint8 HWC in -> uint8 HWC out): %.3f ms per 16 frames = %.0f frames/s" % (d * 1e3, 16 / d), flush=True)
