#!/usr/bin/env python
"""Random-shape parity sweep of ie_mconv_nhwc against torch CPU fp32 on the same 16-bit-rounded operands."""
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
    k = rnd.choice([1, 3, 3, 3, 4, 5, 9])
    s = rnd.choice([1, 1, 1, 2])
    p = rnd.choice([0, k // 2, k // 2, max(0, k // 2 - 1)]) if k > 1 else rnd.choice([0, 1])
    mode = rnd.choice(["zeros", "zeros", "replicate"])
    nsrc = rnd.choice([1, 1, 2, 3])
    cs = [8 * rnd.randint(1, 12) for _ in range(nsrc)]
    if rnd.random() < 0.2:
        cs, nsrc = [8], 1
    co = rnd.choice([1, 3, 8, 16, 24, 32, 40, 64, 72, 128, 136])
    n, h, w = rnd.randint(1, 3), rnd.randint(max(k, 5), 70), rnd.randint(max(k, 5), 90)
    if (h + 2 * p - k) // s + 1 <= 0 or (w + 2 * p - k) // s + 1 <= 0:
        continue
    act = rnd.choice(["none", "relu", "lrelu", "tanh", "sigmoid"])
    out_f32 = rnd.random() < 0.3
    use_res = rnd.random() < 0.3
    clamp = (0.0, 1.0) if rnd.random() < 0.2 else None
    g = torch.Generator().manual_seed(case)
    xs = [torch.randn(n, c, h, w, generator=g).to(dt) for c in cs]
    cin = sum(cs)
    wt = (torch.randn(co, cin, k, k, generator=g) / (cin * k * k) ** 0.5).to(dt).float()
    b = torch.randn(co, generator=g) if rnd.random() < 0.7 else None
    xcat = torch.cat([x.float() for x in xs], 1)
    xp = F.pad(xcat, (p,) * 4, mode="replicate") if mode == "replicate" and p > 0 else xcat
    ref = F.conv2d(xp, wt, b, stride=s, padding=0 if mode == "replicate" else p)
    ref = {"none": lambda t: t, "relu": F.relu, "lrelu": lambda t: F.leaky_relu(t, 0.2), "tanh": torch.tanh, "sigmoid": torch.sigmoid}[act](ref)
    res = None
    if use_res:
        res = torch.randn(ref.shape, generator=g).to(dt)
        ref = ref + res.float()
    if clamp:
        ref = ref.clamp(*clamp)
    mw = ops.MConvWeights(wt, b, cs, dt, "cuda")
    out = ops.mconv2d([x.cuda().contiguous(memory_format=CL) for x in xs], mw, stride=s, pad=p, pad_mode=mode, act=act, slope=0.2,
                      residual=None if res is None else res.cuda().contiguous(memory_format=CL), clamp=clamp, out_f32=out_f32)
    err = float((out.float().cpu() - ref).abs().max() / ref.abs().max().clamp_min(1e-6))
    tol = 3e-5 if out_f32 else (2e-3 if dt == torch.float16 else 1.2e-2)
    ok = tuple(out.shape) == tuple(ref.shape) and err <= tol and bool(torch.isfinite(out).all())
    if not ok:
        bad += 1
        print("FAIL", case, dict(dt=str(dt), k=k, s=s, p=p, mode=mode, cs=cs, co=co, nhw=(n, h, w), act=act, out_f32=out_f32, res=use_res, clamp=clamp), "err", err, flush=True)
print(f"{n_cases} cases, {bad} failures")
