#!/usr/bin/env python
"""Random-shape sweep of the fused Zero-DCE schedule (16-bit, all three storage modes) against the fp32 parity path of the same library:
frame counts around the group size, heights around the row-chunk boundaries, widths around the strip pitches (122 / 124 / 126 / 128).
    python tools/fuzz_zdce.py [seed] [cases]"""
import os, random, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import mon_amd
from oracle import zero_dce as o

rnd = random.Random(int(sys.argv[1]) if len(sys.argv) > 1 else 0)
n_cases = int(sys.argv[2]) if len(sys.argv) > 2 else 40
sd = o.init_state_dict(seed=1, gain=3.0)
models = {dt: mon_amd.MODELS.build(name="zero_dce_re", weights=dict(sd), compute_dtype=dt).to("cuda").eval() for dt in ("f32", "f16", "bf16", "bf16_pure")}
tol = {"f16": 1.5e-3, "bf16": 3e-3, "bf16_pure": 1.2e-2}
bad = 0
for case in range(n_cases):
    n = rnd.choice([1, 2, 3, 5, 15, 16, 17, 33])
    h = rnd.choice([1, 2, 7, 8, 9, 16, 67, 68, 69, 135, rnd.randint(1, 300)])
    w = rnd.choice([1, 7, 121, 122, 123, 124, 125, 126, 127, 128, 129, 244, 245, 248, 249, 252, 253, rnd.randint(1, 500)])
    if n * h * w > 6_000_000:
        n = max(1, 6_000_000 // (h * w))
    g = torch.Generator().manual_seed(case)
    x = (torch.rand(n, 3, h, w, generator=g) ** 2.2 * rnd.choice([0.3, 1.0])).cuda()
    ref = models["f32"].enhance(x, want_enhanced=True, want_u8=True)
    for dt in ("f16", "bf16", "bf16_pure"):
        out = models[dt].enhance(x, want_enhanced=True, want_u8=True)
        err = float((out["enhanced"] - ref["enhanced"]).abs().max() / ref["enhanced"].abs().max().clamp_min(1e-6))
        du8 = int((out["u8"].int() - ref["u8"].int()).abs().max())
        if not (err <= tol[dt]) or du8 > (2 if dt == "bf16_pure" else 1) or not bool(torch.isfinite(out["enhanced"]).all()):
            bad += 1
            print(f"FAIL case {case}: n={n} h={h} w={w} dt={dt} err={err:.3e} u8 diff {du8}")
        # the last kernel's other compiled forms (enhanced alone / enhanced + curve maps / uint8 alone) must give the same bits as the general one
        e_only = models[dt].enhance(x)["enhanced"]
        ea = models[dt].enhance(x, want_adjust=True)
        u_only = models[dt].enhance(x, want_enhanced=False, want_u8=True)["u8"]
        if not (torch.equal(e_only, out["enhanced"]) and torch.equal(ea["enhanced"], out["enhanced"]) and torch.equal(u_only, out["u8"]) and bool(torch.isfinite(ea["adjust"]).all())):
            bad += 1
            print(f"FAIL case {case}: n={n} h={h} w={w} dt={dt}: output-mode forms disagree")
print(f"{n_cases} cases x 3 modes, {bad} failures")
sys.exit(1 if bad else 0)
