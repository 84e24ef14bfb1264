#!/usr/bin/env python
"""Random-shape sweep of ie_mconv_pool_nhwc (conv + bias + act with the MaxPool2d(2) behind it): the full-resolution output must be ie_mconv_nhwc's bit for bit,
the pooled output F.max_pool2d of it bit for bit, whichever kernel took the layer (strip kernel's parked rows, the tiled kernel's pooling epilogue at both tile
heights, or conv + pool kernel); the conv itself is checked against torch CPU fp32 on the same 16-bit operands.  python tools/fuzz_mconv_pool.py [seed] [cases]"""
import os, random, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torch.nn.functional as F
from mon_amd import ops

CL = torch.channels_last
rnd = random.Random(int(sys.argv[1]) if len(sys.argv) > 1 else 0)
n_cases = int(sys.argv[2]) if len(sys.argv) > 2 else 200
g = torch.Generator().manual_seed(rnd.randint(0, 1 << 30))
bad = 0
for case in range(n_cases):
    dt = rnd.choice([torch.float16, torch.bfloat16])
    nsrc = rnd.choice([1, 1, 2, 3])
    cs = [8 * rnd.choice([1, 2, 4, 4, 8, 8, 12, 16, 24, 32]) for _ in range(nsrc)]
    co = rnd.choice([16, 32, 48, 64, 64, 80, 128, 128, 192, 256, 24, 40])
    n, h, w = rnd.randint(1, 3), rnd.choice([2, 3, 7, 8, 9, 15, 16, 17, 31, 33, 48, 67, 70]), rnd.choice([2, 3, 31, 32, 33, 47, 64, 65, 90, 97, 128])
    act = rnd.choice(["relu", "relu", "relu", "none", "lrelu"])
    mode = rnd.choice(["zeros", "zeros", "replicate"])
    xs = [torch.randn(n, c, h, w, generator=g).to(dt).cuda().contiguous(memory_format=CL) for c in cs]
    cin = sum(cs)
    wt = (torch.randn(co, cin, 3, 3, generator=g) / (cin * 9) ** 0.5).to(dt).float()
    b = torch.randn(co, generator=g) * 0.3
    mw = ops.MConvWeights(wt, b, cs, dt, "cuda")
    kw = dict(act=act, pad_mode=mode, slope=0.2) if act == "lrelu" else dict(act=act, pad_mode=mode)
    plain = ops.mconv2d(xs, mw, **kw)
    out, pooled = ops.mconv2d(xs, mw, pool=True, **kw)
    ok = torch.equal(out, plain) and tuple(pooled.shape) == (n, co, h // 2, w // 2) and torch.equal(pooled.float().cpu(), F.max_pool2d(plain.float().cpu(), 2))
    xin = F.pad(torch.cat([x.float().cpu() for x in xs], 1), (1, 1, 1, 1), mode="replicate" if mode == "replicate" else "constant")
    ref = F.conv2d(xin, wt, b)
    ref = {"relu": F.relu, "none": lambda t: t, "lrelu": lambda t: F.leaky_relu(t, 0.2)}[act](ref)
    err = float((plain.float().cpu() - ref).abs().max() / ref.abs().max().clamp_min(1e-6))
    tol = 3e-3 if dt == torch.float16 else 2e-2
    if not ok or not err <= tol:
        bad += 1
        print(f"MISMATCH case {case}: dt={dt} cs={cs} co={co} n={n} h={h} w={w} act={act} pad={mode} pooled_ok={ok} err={err:.2e}", flush=True)
    if case % 50 == 49:
        print(f"... {case + 1} cases, {bad} mismatches", flush=True)
print(f"fuzz_mconv_pool: {n_cases} cases, {bad} mismatches")
sys.exit(1 if bad else 0)
