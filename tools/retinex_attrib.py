#!/usr/bin/env python
"""Which fp16 rounding carries RetinexNet's error?  (1.58e-3 on `enhanced` at 1080p, bound 1e-3.)
CPU emulation of the 16-bit path of generators.RetinexNet (retinexnet/model.py:21-90): every stored tensor and every weight
tensor can be rounded to fp16 / bf16 on its own; prints max |err| / max |ref| of `enhanced` per source and for the shipped schedule.
    python tools/retinex_attrib.py [H W] [f16|bf16]
"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import torch.nn.functional as F

import mon_amd

H, W = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (540, 960)
DT = {"f16": torch.float16, "bf16": torch.bfloat16}[sys.argv[3] if len(sys.argv) > 3 else "f16"]
torch.set_num_threads(os.cpu_count() or 8)
m = mon_amd.MODELS.build(name="retinexnet", seed=1)
sd = m.state_dict()
x = torch.rand(1, 3, H, W, generator=torch.Generator().manual_seed(0))
rnd = lambda t: t.to(DT).to(torch.float32)      # noqa: E731


def forward(act=(), wts=(), r_f32=True):
    """act / wts: names of activation tensors / conv layers whose values / weights are rounded.  r_f32: S = R * I_delta uses the fp32 R."""
    def c(name, t, stride=1, padding=1, valid=False):
        w = sd[f"{name}.weight"]
        w = rnd(w) if (name in wts or "*" in wts) else w
        if padding and not valid:
            t = F.pad(t, (padding,) * 4, mode="replicate")
        return F.conv2d(t, w, sd[f"{name}.bias"], stride=stride)
    q = lambda name, t: rnd(t) if (name in act or "*" in act) else t      # noqa: E731
    xin = q("xin", torch.cat((x.max(1, keepdim=True)[0], x), 1))
    f = q("conv0", c("net1_conv0", xin, padding=4))
    for i in (0, 2, 4, 6, 8):
        f = q(f"convs{i}", F.relu(c(f"net1_convs.{i}", f)))
    rl32 = torch.sigmoid(c("net1_recon", f))
    rl = q("rl", rl32)
    out0 = q("out0", c("net2_conv0_1", rl))
    out1 = q("out1", F.relu(c("net2_conv1_1", out0, stride=2)))
    out2 = q("out2", F.relu(c("net2_conv1_2", out1, stride=2)))
    out3 = q("out3", F.relu(c("net2_conv1_3", out2, stride=2)))
    sz = lambda t: (t.shape[2], t.shape[3])      # noqa: E731
    d1 = q("d1", F.relu(c("net2_deconv1_1", torch.cat((F.interpolate(out3, size=sz(out2)), out2), 1))))
    d2 = q("d2", F.relu(c("net2_deconv1_2", torch.cat((F.interpolate(d1, size=sz(out1)), out1), 1))))
    d3 = q("d3", F.relu(c("net2_deconv1_3", torch.cat((F.interpolate(d2, size=sz(out0)), out0), 1))))
    feats = torch.cat((F.interpolate(d1, size=sz(x)), F.interpolate(d2, size=sz(x)), d3), 1)
    fus = q("fus", c("net2_fusion", feats, padding=1))
    i_delta = F.conv2d(fus, rnd(sd["net2_output.weight"]) if ("net2_output" in wts or "*" in wts) else sd["net2_output.weight"], sd["net2_output.bias"])
    r = rl32[:, :3] if r_f32 else rl[:, :3]
    return r * i_delta, i_delta, rl32


with torch.no_grad():
    ref, idl, _ = forward()
    err = lambda e: float((e - ref).abs().max() / ref.abs().max())      # noqa: E731
    print(f"RetinexNet {H}x{W} {DT}: |enhanced| max {float(ref.abs().max()):.3f}, |I_delta| max {float(idl.abs().max()):.3f}")
    acts = ["xin", "conv0", "convs0", "convs2", "convs4", "convs6", "convs8", "rl", "out0", "out1", "out2", "out3", "d1", "d2", "d3", "fus"]
    layers = ["net1_conv0"] + [f"net1_convs.{i}" for i in (0, 2, 4, 6, 8)] + ["net1_recon", "net2_conv0_1"] + [f"net2_conv1_{i}" for i in (1, 2, 3)] + \
             [f"net2_deconv1_{i}" for i in (1, 2, 3)] + ["net2_fusion", "net2_output"]
    print(f"  shipped r02 (all activations + weights, R from the 16-bit tensor): {err(forward(('*',), ('*',), r_f32=False)[0]):.3e}")
    print(f"  all activations + weights, R in fp32                              : {err(forward(('*',), ('*',), r_f32=True)[0]):.3e}")
    print(f"  all weights only                                                  : {err(forward((), ('*',))[0]):.3e}")
    print(f"  all activations only (R fp32)                                     : {err(forward(('*',), ())[0]):.3e}")
    print(f"  R rounded only                                                    : {err(forward(('rl',), (), r_f32=False)[0]):.3e}")
    for a in acts:
        print(f"    activation {a:8s}: {err(forward((a,), ())[0]):.3e}")
    for l in layers:
        print(f"    weights    {l:16s}: {err(forward((), (l,))[0]):.3e}")
