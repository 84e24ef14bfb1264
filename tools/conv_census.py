#!/usr/bin/env python
"""Per-shape census of the matrix-core conv calls of one model forward at 1080p: calls, total GPU time (HIP events around every
call, so launch gaps are excluded), FLOP rate and algorithmic HBM rate.  Usage: python tools/conv_census.py MODEL [bf16|f16]"""
import collections, json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import mon_amd
from mon_amd import ops

name = sys.argv[1] if len(sys.argv) > 1 else "hinet64"
dt = sys.argv[2] if len(sys.argv) > 2 else "bf16"
CFG = {"hinet64": (dict(name="hinet_re"), 1, (1088, 1920)), "llunet++": (dict(name="llunet++_re"), 1, (1088, 1920)),
       "mprnet": (dict(name="mprnet"), 1, (1080, 1920)), "retinexnet": (dict(name="retinexnet"), 2, (1080, 1920)),
       "unet32": (dict(name="unet", base_channels=32, num_classes=3, enhance=True), 8, (1080, 1920)),
       "unet32_ps": (dict(name="unet", base_channels=32, num_classes=3, enhance=True, upsampler="pixelshuffle"), 8, (1080, 1920)),
       "unet64": (dict(name="unet", base_channels=64, num_classes=3, enhance=True), 8, (1080, 1920)),
       "lllinet": (dict(name="lllinet"), 1, (1080, 1920))}
cfg, frames, hw = CFG[name]
m = mon_amd.MODELS.build(seed=1, compute_dtype=dt, **cfg).to("cuda")
x = torch.rand(frames, 3, *hw, device="cuda")
m.forward({"image": x})
torch.cuda.synchronize()
rec = collections.OrderedDict()
orig = ops.mconv2d


def logged(srcs, mw, stride=1, pad=1, **kw):
    s = [srcs] if isinstance(srcs, torch.Tensor) else list(srcs)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    out = orig(srcs, mw, stride=stride, pad=pad, **kw)
    e1.record()
    e1.synchronize()
    o = out[0] if isinstance(out, tuple) else out
    key = ("+".join(str(int(t.shape[1])) for t in s), mw.cout, f"k{mw.kh}s{stride}p{pad}", f"{int(s[0].shape[2])}x{int(s[0].shape[3])}",
           kw.get("act", "none"), "res" if kw.get("residual") is not None else "", "f32" if kw.get("out_f32") else "")
    r = rec.setdefault(key, [0, 0.0, 0.0, 0.0])
    r[0] += 1
    r[1] += e0.elapsed_time(e1)
    npx = o.shape[0] * o.shape[2] * o.shape[3]
    r[2] += 2.0 * npx * o.shape[1] * sum(int(t.shape[1]) for t in s) * mw.kh * mw.kw
    r[3] += sum(t.numel() * t.element_size() for t in s) + o.numel() * o.element_size()
    return out


ops.mconv2d = logged
m.forward({"image": x})
torch.cuda.synchronize()
tot = sum(r[1] for r in rec.values())
print(f"{name} {dt} frames={frames}: {len(rec)} shapes, {sum(r[0] for r in rec.values())} calls, {tot:.2f} ms in mconv2d")
for key, r in sorted(rec.items(), key=lambda kv: -kv[1][1])[:24]:
    print(f"{r[1]:8.3f} ms {100 * r[1] / tot:5.1f}%  x{r[0]:<3d} {r[2] / r[1] / 1e9:7.0f} TFLOP/s {r[3] / r[1] / 1e9:6.2f} TB/s  " + " ".join(map(str, key)))
