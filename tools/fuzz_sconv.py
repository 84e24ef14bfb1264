#!/usr/bin/env python
"""Random-shape parity sweep of the strip form of the narrow 3x3 layers (ie_conv_strip.hip, reached through ie_mconv_nhwc) against torch
CPU fp32 on the same 16-bit-rounded operands: image sizes around the strip / row-chunk boundaries, every channel / activation / padding
combination the kernel covers.   python tools/fuzz_sconv.py [seed] [cases]"""
import os, random, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torch.nn.functional as F
from mon_amd import ops

CL = torch.channels_last
rnd = random.Random(int(sys.argv[1]) if len(sys.argv) > 1 else 0)
n_cases = int(sys.argv[2]) if len(sys.argv) > 2 else 120
bad = 0
worst = 0.0
for case in range(n_cases):
    dt = rnd.choice([torch.float16, torch.bfloat16])
    cs = rnd.choice([[32], [64], [32, 32], [64, 64], [128]])
    co = 64 if sum(cs) == 128 else rnd.choice([32, 64])
    mode = rnd.choice(["zeros", "zeros", "replicate"])
    act = rnd.choice(["none", "relu", "lrelu"])
    n = rnd.randint(1, 3)
    h = rnd.choice([1, 2, 3, 5, 16, 17, 23, 24, 25, 47, 48, 49, rnd.randint(1, 120)])
    w = rnd.choice([64, 65, 127, 128, 129, 130, 255, 256, 257, rnd.randint(64, 400)])
    g = torch.Generator().manual_seed(1000 + case)
    xs = [torch.randn(n, c, h, w, generator=g).to(dt) for c in cs]
    cin = sum(cs)
    wt = (torch.randn(co, cin, 3, 3, generator=g) / (cin * 9) ** 0.5).to(dt).float()
    b = torch.randn(co, generator=g) if rnd.random() < 0.7 else None
    xcat = torch.cat([x.float() for x in xs], 1)
    xp = F.pad(xcat, (1,) * 4, mode="replicate") if mode == "replicate" else xcat
    ref = F.conv2d(xp, wt, b, padding=0 if mode == "replicate" else 1)
    ref = {"none": lambda t: t, "relu": F.relu, "lrelu": lambda t: F.leaky_relu(t, 0.2)}[act](ref)
    mw = ops.MConvWeights(wt, b, cs, dt, "cuda")
    out = ops.mconv2d([x.cuda().contiguous(memory_format=CL) for x in xs], mw, pad=1, pad_mode=mode, act=act, slope=0.2)
    err = float((out.float().cpu() - ref).abs().max() / ref.abs().max().clamp_min(1e-6))
    tol = 2e-3 if dt == torch.float16 else 1.2e-2
    worst = max(worst, err / tol)
    if not (err <= tol) or tuple(out.shape) != tuple(ref.shape):
        bad += 1
        print(f"FAIL case {case}: dt={dt} cs={cs} co={co} n={n} h={h} w={w} mode={mode} act={act} bias={b is not None} err={err:.3e}")
print(f"{n_cases} cases, {bad} failures, worst err / tol {worst:.3f}")
sys.exit(1 if bad else 0)
