#!/usr/bin/env python
"""Sweep the plan's schedule knobs (rows per workgroup, band rows, frames per group) at 1080p and
print per-layer kernel time (hipEvents on the launch stream).  Usage: tools/tune.py [N] [dtype]"""
import ctypes as C
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch

import mon_amd
from mon_amd import capi
from oracle import zero_dce as o_zdce

N = int(sys.argv[1]) if len(sys.argv) > 1 else 16
dtype = sys.argv[2] if len(sys.argv) > 2 else "bf16"
sd = o_zdce.init_state_dict(seed=1, gain=3.0)
m = mon_amd.MODELS.build(name="zero_dce_re", weights=dict(sd), compute_dtype=dtype).to("cuda")
x = torch.rand(N, 3, 1080, 1920, device="cuda")
out = {"enhanced": torch.empty_like(x)}
L = capi.lib()


def run(rb, band, group, reps=3):
    m.set_tuning(rb, band, group)
    for _ in range(2):
        m.enhance(x, out=out)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        m.enhance(x, out=out)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / reps
    plan = m._ensure_plan()
    ms = (C.c_float * 8)(); names = (C.c_char_p * 8)(); launches = (C.c_int * 8)(); cnt = C.c_int(0)
    capi.check(L.ie_zdce_profile_forward(plan, capi.ptr(x), capi.ptr(out["enhanced"]), None, None, N, 1080, 1920,
                                         capi.current_stream(), ms, names, launches, 8, C.byref(cnt)))
    print(f"rb={rb:3d} band={band:4d} group={group:2d}: {dt*1e3:7.2f} ms  {N/dt:7.1f} img/s | " +
          " ".join(f"{ms[i]:.2f}" for i in range(7)) + f" | launches {launches[1]}", flush=True)


CFGS = [(0, 0, 0), (135, 0, 8), (270, 0, 8), (540, 0, 8), (1080, 0, 8), (270, 0, 16), (540, 0, 16), (1080, 0, 16), (270, 0, 4), (135, 0, 4), (540, 0, 32), (1080, 0, 32)]
if len(sys.argv) > 3 and sys.argv[3] == "bands":       # row bands small enough for the intermediates of a band to stay in the 256 MB Infinity Cache
    CFGS = [(0, 0, 0), (0, 540, 8), (0, 270, 8), (0, 128, 8), (0, 64, 8), (0, 270, 4), (0, 128, 4), (0, 270, 16), (0, 128, 16), (64, 128, 8), (32, 64, 8)]
if len(sys.argv) > 3 and sys.argv[3] == "mall":        # small frame groups / row bands: do x2 / x5 of one job survive in the Infinity Cache between the kernels? (the schedule is power-capped)
    CFGS = [(0, 0, 0), (0, 0, 2), (0, 0, 4), (0, 0, 8), (0, 0, 0), (0, 270, 16), (0, 136, 16), (0, 68, 16), (0, 0, 0), (0, 540, 4), (0, 270, 8), (0, 540, 8)]
for cfg in CFGS:
    if cfg[2] > N:
        continue
    run(*cfg)
