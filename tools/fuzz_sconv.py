#!/usr/bin/env python
"""Random-shape parity sweep of the strip form of the narrow 3x3 layers (ie_conv_strip.hip, reached through ie_mconv_nhwc) against torch
CPU fp32 on the same 16-bit-rounded operands: image sizes around the strip / row-chunk boundaries, every channel / activation / padding
combination the kernel covers, stride 2, and the fused forms on the rows parked in LDS (2x2 max-pool: bit-exact against F.max_pool2d of
the plain launch; 1x1 tail: against the two-launch form).   python tools/fuzz_sconv.py [seed] [cases]"""
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
    stride = rnd.choice([1, 1, 1, 2])
    ref = F.conv2d(xp, wt, b, stride=stride, padding=0 if mode == "replicate" else 1)
    ref = {"none": lambda t: t, "relu": F.relu, "lrelu": lambda t: F.leaky_relu(t, 0.2)}[act](ref)
    mw = ops.MConvWeights(wt, b, cs, dt, "cuda")
    xd = [x.cuda().contiguous(memory_format=CL) for x in xs]
    out = ops.mconv2d(xd, mw, stride=stride, pad=1, pad_mode=mode, act=act, slope=0.2)
    if stride == 1 and h >= 2:                                  # fused 2x2 max-pool (strip kernel for ReLU layers, conv + pool kernel otherwise)
        o2, pooled = ops.mconv2d(xd, mw, pad=1, pad_mode=mode, act=act, slope=0.2, pool=True)
        if not (torch.equal(o2, out) and torch.equal(pooled.float().cpu(), F.max_pool2d(out.float().cpu(), 2))):
            bad += 1
            print(f"FAIL pool case {case}: dt={dt} cs={cs} co={co} n={n} h={h} w={w} mode={mode} act={act}")
    if stride == 1 and act == "relu" and w % 4 == 0:            # fused 1x1 tail
        ct = rnd.choice([1, 3, 16])
        w1 = (torch.randn(ct, co, 1, 1, generator=g) / co ** 0.5).to(dt).float()
        b1 = torch.randn(ct, generator=g) * 0.1
        tail = ops.MTailWeights(w1, b1, dt, "cuda")
        if ops.mconv2d_tail_applicable(xd, mw, tail, pad_mode=mode):
            got = ops.mconv2d_tail(xd, mw, tail, pad_mode=mode, clamp=(0.0, 1.0) if ct == 3 else None)
            want = ops.to_nchw(ops.mconv2d(out, ops.MConvWeights(w1, b1, [co], dt, "cuda"), pad=0, clamp=(0.0, 1.0) if ct == 3 else None, out_f32=True))
            e2 = float((got - want).abs().max() / want.abs().max().clamp_min(1e-6))
            if not e2 <= 2e-6:
                bad += 1
                print(f"FAIL tail case {case}: dt={dt} cs={cs} co={co} n={n} h={h} w={w} mode={mode} ct={ct} err={e2:.3e}")
    err = float((out.float().cpu() - ref).abs().max() / ref.abs().max().clamp_min(1e-6))
    tol = 2e-3 if dt == torch.float16 else 1.2e-2
    worst = max(worst, err / tol)
    if not (err <= tol) or tuple(out.shape) != tuple(ref.shape):
        bad += 1
        print(f"FAIL case {case}: dt={dt} cs={cs} co={co} n={n} h={h} w={w} mode={mode} act={act} stride={stride} bias={b is not None} err={err:.3e}")
# the first DoubleConv (+ pool) in one launch against thin conv + strip conv (+ fused pool): bit for bit
for case in range(n_cases // 4):
    dt = rnd.choice([torch.float16, torch.bfloat16])
    c = rnd.choice([32, 64])
    n = rnd.randint(1, 3)
    h = rnd.choice([2, 3, 5, 16, 17, 23, 24, 25, 47, 48, 49, rnd.randint(2, 120)])
    w = rnd.choice([64, 65, 127, 128, 129, 130, 255, 256, 257, rnd.randint(64, 400)])
    g = torch.Generator().manual_seed(5000 + case)
    x8 = ops.to_nhwc16(torch.rand(n, 3, h, w, generator=g).cuda(), dt, 8)
    mw1 = ops.MConvWeights((torch.randn(c, 3, 3, 3, generator=g) / 27 ** 0.5), torch.randn(c, generator=g) * 0.2 if rnd.random() < 0.7 else None, [8], dt, "cuda",
                           cin_map=[0, 1, 2, -1, -1, -1, -1, -1])
    mw2 = ops.MConvWeights((torch.randn(c, c, 3, 3, generator=g) / (c * 9) ** 0.5), torch.randn(c, generator=g) * 0.2, [c], dt, "cuda")
    out, pooled = ops.mconv2d_first_pool(x8, mw1, mw2)
    want, want_p = ops.mconv2d(ops.mconv2d(x8, mw1, act="relu"), mw2, act="relu", pool=True)
    if not (torch.equal(out, want) and torch.equal(pooled, want_p)):
        bad += 1
        print(f"FAIL first case {case}: dt={dt} c={c} n={n} h={h} w={w}")
print(f"{n_cases} cases, {bad} failures, worst err / tol {worst:.3f}")
sys.exit(1 if bad else 0)
