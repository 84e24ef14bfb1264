#!/usr/bin/env python
"""Random-shape parity sweep of the fp32 conv (direct + LDS-tiled 4x4 / 8x4 / 8x8 kernels), its data and weight gradients."""
import os, random, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torch.nn.functional as F
from mon_amd import ops

CL = torch.channels_last
rnd = random.Random(int(sys.argv[1]) if len(sys.argv) > 1 else 0)
n_cases = int(sys.argv[2]) if len(sys.argv) > 2 else 30
bad = 0
for case in range(n_cases):
    big = case % 2 == 0                                    # big: >= 128K output pixels -> the 8x4 / 8x8 tiles
    k = rnd.choice([1, 3, 3, 3, 4])
    s = rnd.choice([1, 1, 1, 2])
    p = k // 2 if k > 1 else 0
    mode = rnd.choice(["zeros", "zeros", "replicate"])
    nsrc = rnd.choice([1, 1, 2, 3])
    cs = [rnd.choice([3, 8, 12, 16, 20, 32, 40]) for _ in range(nsrc)]
    co = rnd.choice([3, 16, 24, 64, 72, 128, 136])
    n = rnd.randint(1, 2)
    h, w = (rnd.randint(260, 300), rnd.randint(260, 300)) if big else (rnd.randint(9, 60), rnd.randint(9, 70))
    if big and s == 2:
        h, w = 2 * h, 2 * w
        if co > 64: co = 64
    act = rnd.choice(["none", "relu", "lrelu"])
    g = torch.Generator().manual_seed(100 + case)
    xs = [torch.randn(n, c, h, w, generator=g).requires_grad_(mode == "zeros") for c in cs]
    cin = sum(cs)
    wt = (torch.randn(co, cin, k, k, generator=g) / (cin * k * k) ** 0.5).requires_grad_(True)
    b = torch.randn(co, generator=g)
    xcat = torch.cat(xs, 1)
    xp = F.pad(xcat, (p,) * 4, mode="replicate") if mode == "replicate" and p > 0 else xcat
    pre = F.conv2d(xp, wt, b, stride=s, padding=0 if mode == "replicate" else p)
    ref = {"none": lambda t: t, "relu": F.relu, "lrelu": lambda t: F.leaky_relu(t, 0.2)}[act](pre)
    cw = ops.ConvWeights(wt.detach(), b, "cuda")
    xd = [ops.to_nhwc(x.detach().cuda()) for x in xs]
    out = ops.conv2d(xd, cw, stride=s, pad=p, pad_mode=mode, act=act, slope=0.2)
    err = float((ops.to_nchw(out).cpu() - ref.detach()).abs().max() / ref.detach().abs().max())
    ok = err < 2e-5
    msg = f"fwd {err:.1e}"
    if mode == "zeros":
        dy = torch.randn(pre.shape, generator=g)
        pre.backward(dy)
        dyd = ops.to_nhwc(dy.cuda())
        dsts = [torch.empty_like(x, memory_format=CL) for x in xd]
        ops.conv2d_dgrad(dyd, cw, [tuple(x.shape) for x in xs], dsts, [0] * len(xs), s, p)
        for d, x in zip(dsts, xs):
            e = float((ops.to_nchw(d).cpu() - x.grad).abs().max() / x.grad.abs().max().clamp_min(1e-9))
            ok = ok and e < 5e-5
            msg += f" dgrad {e:.1e}"
        dw = torch.empty(k * k * cin * co, device="cuda")
        ops.conv2d_wgrad(xd, dyd, cw, dw, None, s, p)
        e = float((dw.reshape(k, k, cin, co).permute(3, 2, 0, 1).cpu() - wt.grad).abs().max() / wt.grad.abs().max())
        ok = ok and e < 2e-4
        msg += f" wgrad {e:.1e}"
    if not ok:
        bad += 1
        print("FAIL", case, dict(k=k, s=s, p=p, mode=mode, cs=cs, co=co, nhw=(n, h, w), act=act), msg, flush=True)
print(f"{n_cases} cases, {bad} failures")
