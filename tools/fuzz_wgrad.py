#!/usr/bin/env python
"""Random-shape parity sweep of the 16-bit weight gradient against torch CPU float64 autograd on the same 16-bit-rounded operands: the matrix-core kernel
for channel counts that are multiples of 8 on 3x3 (pad 0 / 1, stride 1 / 2), 1x1, 4x4 stride 2 pad 1 and 9x9 pad 0 layers (the blocked / parity-sub-image
launches of round 3), the FMA kernel otherwise."""
import os, random, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torch.nn.functional as F
from mon_amd import ops

CL = torch.channels_last
rnd = random.Random(int(sys.argv[1]) if len(sys.argv) > 1 else 0)
n_cases = int(sys.argv[2]) if len(sys.argv) > 2 else 60
bad = 0
for case in range(n_cases):
    dt = rnd.choice([torch.float16, torch.bfloat16])
    k, p, st = rnd.choice([(3, 1, 1), (3, 1, 1), (3, 1, 1), (1, 0, 1), (3, 0, 1), (3, 0, 1), (9, 0, 1), (4, 1, 2), (4, 1, 2), (3, 0, 2), (3, 1, 2), (5, 2, 1)])
    nsrc = rnd.choice([1, 1, 2, 3])
    mult = rnd.choice([8, 8, 8, 4])                      # 4: not every count is a multiple of 8 -> FMA kernel
    cs = [mult * rnd.randint(1, 20) for _ in range(nsrc)]
    co = mult * rnd.randint(1, 20)
    n, h, w = rnd.randint(1, 3), rnd.randint(max(3, k), 80), rnd.randint(max(3, k), 140)
    if k == 9:
        cs, co = [c if c <= 64 else 64 for c in cs], min(co, 96)          # keep the float64 reference quick
    g = torch.Generator().manual_seed(1000 + case)
    xs = [torch.randn(n, c, h, w, generator=g).to(dt) for c in cs]
    cin = sum(cs)
    wt = torch.zeros(co, cin, k, k, dtype=torch.float64, requires_grad=True)
    y = F.conv2d(torch.cat([x.double() for x in xs], 1), wt, stride=st, padding=p)
    dy = torch.randn(y.shape, generator=g).to(dt)
    y.backward(dy.double())
    cw = ops.ConvWeightsView(None, None, co, cin, k, k)
    dw = torch.full((k * k * cin * co,), 3.0, device="cuda")
    ops.conv2d_wgrad([x.cuda().contiguous(memory_format=CL) for x in xs], dy.cuda().contiguous(memory_format=CL), cw, dw, None, st, p)
    got = dw.reshape(k, k, cin, co).permute(3, 2, 0, 1).double().cpu()
    err = float((got - wt.grad).abs().max() / wt.grad.abs().max().clamp_min(1e-30))
    if not (err < 3e-5):
        bad += 1
        print("FAIL", case, dict(dt=str(dt), k=k, p=p, s=st, cs=cs, co=co, nhw=(n, h, w)), "err", err, flush=True)
print(f"{n_cases} cases, {bad} failures")
