#!/usr/bin/env python
"""Where does ie_mconv_pair_nhwc differ from two ie_mconv_nhwc calls?  python tools/pair_diag.py N H W [zeros|replicate] [f16|bf16]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from mon_amd import ops
CL = torch.channels_last
n, h, w = (int(v) for v in sys.argv[1:4])
pm = sys.argv[4] if len(sys.argv) > 4 else "zeros"
tdt = ops.TORCH_DT[sys.argv[5] if len(sys.argv) > 5 else "f16"]
g = torch.Generator().manual_seed(31)
x = torch.randn(n, 64, h, w, generator=g).to(tdt).cuda().contiguous(memory_format=CL)
mode = sys.argv[6] if len(sys.argv) > 6 else "both"          # ida: layer a = identity (tests stage B's arithmetic); idb: layer b = identity (stage A + hand-off)
ws = [(torch.randn(64, 64, 3, 3, generator=g) / 24.0).to(tdt).float() for _ in range(2)]
bs = [torch.randn(64, generator=g) * 0.2 for _ in range(2)]
ident = torch.zeros(64, 64, 3, 3); ident[torch.arange(64), torch.arange(64), 1, 1] = 1.0
if mode == "ida": ws[0], bs[0] = ident, torch.zeros(64)
if mode == "idb": ws[1], bs[1] = ident, torch.zeros(64)
mws = [ops.MConvWeights(ws[i], bs[i], [64], tdt, "cuda") for i in range(2)]
got = ops.mconv2d_pair(x, mws[0], mws[1], pad_mode=pm).float().cpu()
want = ops.mconv2d(ops.mconv2d(x, mws[0], pad_mode=pm, act="relu"), mws[1], pad_mode=pm, act="relu").float().cpu()
bad = (got != want)
print("mismatching elements", int(bad.sum()), "of", bad.numel(), "max abs diff", float((got - want).abs().max()))
if bad.any():
    px = bad.any(1)          # (n, h, w)
    print("frames", sorted(set(px.nonzero()[:, 0].tolist())))
    rows = px.any(2).any(0).nonzero().flatten().tolist(); cols = px.any(1).any(0).nonzero().flatten().tolist()
    print("rows", rows[:40], "..." if len(rows) > 40 else "")
    print("cols", cols[:60], "..." if len(cols) > 60 else "")
    ch = bad.any(3).any(2).any(0).nonzero().flatten().tolist()
    print("channels", ch)
    frac_by_col = px.float().mean((0, 1))
    print("fraction of rows bad per col (first 20 bad cols):", [(c, round(float(frac_by_col[c]), 2)) for c in cols[:20]])
if bad.any():
    y = 5
    for xx in (1, 5, 13, 14, 15, 112, 113, 120):
        if xx < w:
            gv, wv_ = got[0, :8, y, xx], want[0, :8, y, xx]
            # is got[x] equal to want at some other column of the row?
            match = [c for c in range(w) if torch.equal(got[0, :, y, xx], want[0, :, y, c])]
            print(f"x={xx}: got {[round(float(v), 3) for v in gv]} want {[round(float(v), 3) for v in wv_]} got == want at columns {match[:5]}")
