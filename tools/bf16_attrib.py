#!/usr/bin/env python
"""Per-source attribution of the 16-bit storage error of the fused Zero-DCE schedule, on the CPU.

Emulates the shipped schedule's roundings in torch fp32 (the arithmetic between roundings is fp32, like the
MFMA's fp32 accumulate): weights of layer k rounded to a 16-bit format, activations x1..x6 and the partial sum
P = W7a*x1 + b7 rounded where the schedule stores them (x1/x4/x6 in LDS, x2/x3/x5/P in HBM).  For each source
of rounding it reports max|enhanced - fp32 oracle| / max|oracle| (the bench's `parity.max_rel_err`) with ONLY
that source active and with ALL BUT that source active, so the table shows which roundings carry the error.

    python tools/bf16_attrib.py [--h 540 --w 960] [--fmt bf16]
"""
import argparse
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import torch.nn.functional as F
from oracle import zero_dce as o


def rnd(t, fmt):
    if fmt == "f32":
        return t
    if fmt == "bf16":
        return t.to(torch.bfloat16).to(torch.float32)
    if fmt == "f16":
        return t.to(torch.float16).to(torch.float32)
    if fmt == "bf16x2":                       # hi + lo pair of bf16: 16 significant bits
        hi = t.to(torch.bfloat16).to(torch.float32)
        return hi + (t - hi).to(torch.bfloat16).to(torch.float32)
    raise ValueError(fmt)


def emulate(sd, x, fmt):
    """fmt: dict source -> format; sources w1..w7, x1..x6, P, img."""
    def f(k):
        return fmt.get(k, "f32")

    def conv(k, t, w=None, b=True, half=""):
        wk = rnd(sd[f"e_conv{k}.weight"] if w is None else w, f(f"w{k}{half}"))
        return F.conv2d(t, wk, sd[f"e_conv{k}.bias"] if b else None, padding=1)
    xi = rnd(x, f("img"))
    x1 = rnd(F.relu(conv(1, xi)), f("x1"))
    x2 = rnd(F.relu(conv(2, x1)), f("x2"))
    w7 = sd["e_conv7.weight"]
    P = rnd(conv(7, x1, w=w7[:, :32], half="a"), f("P"))
    x3 = rnd(F.relu(conv(3, x2)), f("x3"))
    x4 = rnd(F.relu(conv(4, x3)), f("x4"))
    x5 = rnd(F.relu(conv(5, torch.cat([x3, x4], 1))), f("x5"))
    w6 = sd["e_conv6.weight"]
    x6 = rnd(F.relu(conv(6, x2, w=w6[:, :32], half="a") + conv(6, x5, w=w6[:, 32:], b=False, half="b")), f("x6"))
    a7 = P + conv(7, x6, w=w7[:, 32:], b=False, half="b")
    return o.curve(x, torch.tanh(a7)), a7


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--h", type=int, default=540)
    ap.add_argument("--w", type=int, default=960)
    ap.add_argument("--fmt", default="bf16")
    ap.add_argument("--gain", type=float, default=3.0)
    ap.add_argument("--lowlight", action="store_true")
    ap.add_argument("--quick", action="store_true", help="only the ALL-sources line")
    ap.add_argument("--golden", default="", help="path of a tests/golden/zero_dce_*.npz: its weights and image instead of the seeded ones")
    ap.add_argument("--mix", default="", help="comma list source=fmt overriding --fmt, e.g. P=f16,x6=f16,w7=f16")
    args = ap.parse_args()
    torch.set_num_threads(8)
    sd = o.init_state_dict(seed=1, gain=args.gain)
    x = torch.rand(1, 3, args.h, args.w, generator=torch.Generator().manual_seed(1234))
    if args.lowlight:
        x = x ** 2.2 * 0.25
    if args.golden:
        from tests.helpers import load_golden
        sd, gd = load_golden(args.golden)
        x = torch.from_numpy(gd["image"])
    with torch.no_grad():
        ref = o.forward(sd, x, return_features=True)
        refe = ref["enhanced"]
        scale = float(refe.abs().max())
        srcs = [f"w{k}" for k in range(1, 6)] + ["w6a", "w6b", "w7a", "w7b"] + [f"x{k}" for k in range(1, 7)] + ["P"]
        allf = {s: args.fmt for s in srcs}
        for kv in filter(None, args.mix.split(",")):
            k, v = kv.split("=")
            allf[k] = v

        def err(fm):
            e, _ = emulate(sd, x, fm)
            d = (e - refe).abs()
            return float(d.max()) / scale, float((d.double() ** 2).mean().sqrt()) / scale
        base = err(allf)
        print(f"size {args.h}x{args.w} fmt {args.fmt} mix [{args.mix}] gain {args.gain}: ALL sources -> max_rel {base[0]:.3e} rms_rel {base[1]:.3e}")
        if args.quick:
            return
        print(f"{'source':8s} {'only(max)':>11s} {'only(rms)':>11s} {'all-but(max)':>13s} {'all-but(rms)':>13s}")
        for s in srcs:
            only = err({s: allf[s]})
            but = err({k: v for k, v in allf.items() if k != s})
            print(f"{s:8s} {only[0]:11.3e} {only[1]:11.3e} {but[0]:13.3e} {but[1]:13.3e}")
        for grp, name in (([k for k in srcs if k[0] == "w"], "weights"), (["x%d" % k for k in range(1, 7)] + ["P"], "activ+P")):
            only = err({k: allf[k] for k in grp})
            print(f"{name:8s} {only[0]:11.3e} {only[1]:11.3e}")


if __name__ == "__main__":
    main()
