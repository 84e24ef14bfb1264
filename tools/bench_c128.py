import sys, time; sys.path.insert(0, "/root/repo")
import torch
from mon_amd import ops
CL = torch.channels_last
g = torch.Generator().manual_seed(1)
n, h, w = int(sys.argv[1]), 1080, 1920
a = torch.randn(n, 64, h // 2, w // 2, generator=g).half().cuda().contiguous(memory_format=CL)
b = torch.randn(n, 64, h, w, generator=g).half().cuda().contiguous(memory_format=CL)
wt = (torch.randn(64, 128, 3, 3, generator=g) / 34.0)
mw = ops.MConvWeights(wt, torch.randn(64, generator=g), [64, 64], torch.float16, "cuda")
for _ in range(5): ops.mconv2d([a, b], mw, pad_mode="replicate", act="relu")
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(20): ops.mconv2d([a, b], mw, pad_mode="replicate", act="relu")
torch.cuda.synchronize(); d = (time.perf_counter() - t0) / 20
print("c128_o64 (nearest x2 + skip) %d frames: %.1f us = %.0f TFLOP/s" % (n, d * 1e6, n * h * w * 128 * 64 * 9 * 2 / d / 1e12), flush=True)
