#!/usr/bin/env python
"""Randomised shapes through the strip kernels added in round 3, each against the launches it replaces (or torch on the same 16-bit inputs):
ie_unet_first_pool_nchw, ie_mconv_pair_nhwc, ie_mconv_pair_thin_nhwc, ie_conv9x9_c4_nhwc, the thin-output form of ie_mconv_nhwc, ie_unet_up_head_nhwc (both forms).
Usage: python tools/fuzz_round3.py [seconds] [seed]"""
import os, sys, time, random
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torch.nn.functional as F
from mon_amd import ops

CL = torch.channels_last
budget = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
rng = random.Random(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
g = torch.Generator().manual_seed(rng.randrange(1 << 30))


def rel(a, b):
    return float((a.double() - b.double()).abs().max() / b.double().abs().max().clamp_min(1e-30))


def t16(n, c, h, w, tdt):
    return torch.randn(n, c, h, w, generator=g).to(tdt).cuda().contiguous(memory_format=CL)


def mw(cout, split, tdt, scale=None):
    cin = sum(split)
    w = (torch.randn(cout, cin, 3, 3, generator=g) / (scale or (cin * 9) ** 0.5)).to(tdt).float()
    return ops.MConvWeights(w, torch.randn(cout, generator=g) * 0.2, list(split), tdt, "cuda"), w


def case_pair(tdt, n, h, w, pm):
    x = t16(n, 64, h, w, tdt)
    a, _ = mw(64, [64], tdt); b, _ = mw(64, [64], tdt)
    got = ops.mconv2d_pair(x, a, b, pad_mode=pm)
    want = ops.mconv2d(ops.mconv2d(x, a, pad_mode=pm, act="relu"), b, pad_mode=pm, act="relu")
    return 0.0 if torch.equal(got, want) else 1.0, 0.0


def case_pair_pool(tdt, n, h, w, pm):
    h = max(h, 2)
    x = t16(n, 32, h, w, tdt)
    a, _ = mw(64, [32], tdt); b, _ = mw(64, [64], tdt)
    y, yp = ops.mconv2d_pair_pool(x, a, b, pad_mode=pm)
    want, want_p = ops.mconv2d(ops.mconv2d(x, a, pad_mode=pm, act="relu"), b, pad_mode=pm, act="relu", pool=True)
    return 0.0 if (torch.equal(y, want) and torch.equal(yp, want_p)) else 1.0, 0.0


def case_pair_thin(tdt, n, h, w, pm):
    x = t16(n, 64, h, w, tdt)
    cout = rng.choice([4, 8, 12, 16]); act = rng.choice(["none", "relu", "sigmoid", "tanh"])
    a, _ = mw(64, [64], tdt); b, _ = mw(cout, [64], tdt)
    got = ops.mconv2d_pair_thin(x, a, b, pad_mode=pm, act_b=act, copy16=cout == 4)
    got = got[0] if cout == 4 else got
    want = ops.mconv2d(ops.mconv2d(x, a, pad_mode=pm, act="relu"), b, pad_mode=pm, act=act, out_f32=True)
    return rel(got, want), 3e-6


def case_conv9(tdt, n, h, w, pm):
    img = torch.rand(n, 3, h, w, generator=g)
    x4 = ops.retinex_decom_input(img.cuda(), tdt, 4)
    wt = (torch.randn(64, 4, 9, 9, generator=g) / 18.0).to(tdt).float(); bias = torch.randn(64, generator=g) * 0.2
    got = ops.conv9x9_c4(x4, ops.Conv9x9C4Weights(wt, bias, tdt, "cuda"))
    ref = F.conv2d(F.pad(x4.float().cpu(), (4, 4, 4, 4), mode="replicate"), wt, bias)
    return rel(got.float().cpu(), ref), 1e-3 if tdt == torch.float16 else 8e-3


def case_thin3(tdt, n, h, w, pm):
    scales = rng.choice([(1,), (2, 1), (4, 2, 1), (1, 1, 1), (3, 1), (4, 1, 2)])
    cout = rng.randint(1, 4); act = rng.choice(["none", "relu", "sigmoid", "tanh"])
    srcs = [t16(n, 64, max(h // sc, 1), max(w // sc, 1), tdt) for sc in scales]
    m, wt = mw(cout, [64] * len(scales), tdt)
    got = ops.mconv2d(srcs, m, pad_mode=pm, act=act, out_f32=True)
    up = torch.cat([F.interpolate(t.float().cpu(), size=(h, w), mode="nearest") for t in srcs], 1)
    ref = F.conv2d(F.pad(up, (1, 1, 1, 1), mode="replicate" if pm == "replicate" else "constant"), wt, m.b.cpu())
    ref = {"sigmoid": torch.sigmoid, "tanh": torch.tanh, "relu": F.relu, "none": lambda t: t}[act](ref)
    return rel(got.cpu(), ref), 3e-5


def case_updc(tdt, n, h, w, pm):
    half = rng.random() < 0.5 and h >= 4
    if half:
        h -= h & 1; w -= w & 1
    classes = rng.randint(1, 16)
    skip = t16(n, 32, h, w, tdt)
    low = t16(n, 32, h // 2, w // 2, tdt) if half else None
    up = ops.upsample2x_bilinear(low) if half else t16(n, 32, h, w, tdt)
    a, _ = mw(32, [32, 32], tdt, 24.0); b, _ = mw(32, [32], tdt, 17.0)
    w1 = (torch.randn(classes, 32, 1, 1, generator=g) / 32 ** 0.5).to(tdt).float(); b1 = torch.randn(classes, generator=g) * 0.1
    tail = ops.MTailWeights(w1, b1, tdt, "cuda")
    clamp = rng.choice([None, (0.0, 1.0)])
    if not ops.unet_up_head_applicable(skip, low if half else up, a, b, tail):
        return 0.0, 0.0
    got = ops.unet_up_head(skip, low if half else up, a, b, tail, clamp=clamp)
    x2 = ops.mconv2d(ops.mconv2d([skip, up], a, act="relu"), b, act="relu")
    want = ops.to_nchw(ops.mconv2d(x2, ops.MConvWeights(w1, b1, [32], tdt, "cuda"), pad=0, clamp=clamp, out_f32=True))
    return rel(got, want), (2e-3 if tdt == torch.float16 else 2e-2) if half else 3e-6      # half: a rare 1-ulp flip of an up-sampled value, amplified by the two convs


def case_first(tdt, n, h, w, pm):
    img = torch.rand(n, 3, h, w, generator=g)
    w1 = (torch.randn(32, 3, 3, 3, generator=g) / 27 ** 0.5).to(tdt).float(); w2 = (torch.randn(32, 32, 3, 3, generator=g) / 17.0).to(tdt).float()
    b1, b2 = torch.randn(32, generator=g) * 0.2, torch.randn(32, generator=g) * 0.2
    fw = ops.UNetFirstWeights(w1, b1, tdt, "cuda"); m2 = ops.MConvWeights(w2, b2, [32], tdt, "cuda")
    if h < 2 or not ops.unet_first_pool_applicable(img.cuda(), fw, m2):
        return 0.0, 0.0
    out, pooled = ops.unet_first_pool(img.cuda(), fw, m2)
    if not torch.equal(pooled.float().cpu(), F.max_pool2d(out.float().cpu(), 2)):
        return 1.0, 0.0
    ref = F.relu(F.conv2d(F.relu(F.conv2d(img.to(tdt).float(), w1, b1, padding=1)).to(tdt).float(), w2, b2, padding=1))
    return rel(out.float().cpu(), ref), 3e-3 if tdt == torch.float16 else 2e-2


CASES = {"first": case_first, "pair": case_pair, "pair_pool": case_pair_pool, "pair_thin": case_pair_thin, "conv9": case_conv9, "thin3": case_thin3, "updc": case_updc}
t0, done, bad = time.time(), {k: 0 for k in CASES}, 0
while time.time() - t0 < budget:
    name = rng.choice(list(CASES))
    tdt = rng.choice([torch.float16, torch.bfloat16])
    n, h = rng.randint(1, 3), rng.choice([1, 2, 3, 5, 8, 13, 31, 64, 97, 130])
    w = rng.choice([64, 65, 100, 124, 125, 126, 127, 128, 129, 130, 190, 248, 250, 252, 256, 300, 379])
    pm = rng.choice(["zeros", "replicate"]) if name != "updc" else "zeros"
    err, tol = CASES[name](tdt, n, h, w, pm)
    done[name] += 1
    if not err <= tol:
        bad += 1
        print(f"MISMATCH {name} dtype={tdt} n={n} h={h} w={w} pad={pm}: err {err:.3e} > {tol:.1e}", flush=True)
torch.cuda.synchronize()
print(f"fuzz_round3: {sum(done.values())} cases {done}, {bad} mismatches, {time.time() - t0:.0f} s", flush=True)
sys.exit(1 if bad else 0)
