#!/usr/bin/env python
"""First-contact diagnostics on the GPU box: runs the Zero-DCE path in every dtype/variant on the
goldens and prints per-tensor errors (does not assert).  Output -> stdout and gpurun_out/diag.json."""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch

import mon_amd
from mon_amd import capi
from oracle import zero_dce as o_zdce
from tests.helpers import load_golden, psnr, rel_err, zero_dce_goldens

res = {"device": capi.device_info(0), "cases": []}
print(res["device"])
for path in zero_dce_goldens():
    sd, g = load_golden(path)
    x = torch.from_numpy(g["image"])
    feats = o_zdce.forward(sd, x, return_features=True)["_features"]
    for dtype in ("f32", "f16", "bf16"):
        for variant in ((0,) if dtype == "f32" else (0, 8)):
            try:
                m = mon_amd.MODELS.build(name="zero_dce_re", weights=dict(sd), compute_dtype=dtype, variant=variant).to("cuda")
                out = m.forward({"image": x.cuda()})
                torch.cuda.synchronize()
                u8 = m.enhance(x.cuda(), want_enhanced=False, want_u8=True)["u8"].cpu().numpy()
                d = np.abs(u8.astype(int) - g["enhanced_u8"].astype(int))
                rec = dict(case=os.path.basename(path)[9:-4], dtype=dtype, variant=variant,
                           enh=rel_err(out["enhanced"].cpu(), g["enhanced"]), adj=rel_err(out["adjust"].cpu(), g["adjust"]),
                           psnr=psnr(out["enhanced"].cpu(), g["enhanced"]), u8_max=int(d.max()), u8_mismatch=float((d != 0).mean()),
                           feats=[rel_err(m.feature(k).cpu(), feats[k - 1]) for k in ((2, 3, 5) if (variant == 0 and dtype != 'f32') else range(1, 7))])
            except Exception as e:  # noqa
                rec = dict(case=os.path.basename(path)[9:-4], dtype=dtype, variant=variant, error=repr(e))
            res["cases"].append(rec)
            print(json.dumps(rec))
os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
json.dump(res, open(os.path.join(ROOT, "gpurun_out", "diag.json"), "w"), indent=1)

# quick timing at 1080p
sd = o_zdce.init_state_dict(seed=1, gain=3.0)
for dtype in ("bf16", "f16"):
    for n in (8,):
        m = mon_amd.MODELS.build(name="zero_dce_re", weights=dict(sd), compute_dtype=dtype).to("cuda")
        x = torch.rand(n, 3, 1080, 1920, device="cuda")
        out = {"enhanced": torch.empty_like(x)}
        for _ in range(2):
            m.enhance(x, out=out)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(5):
            m.enhance(x, out=out)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / 5
        print(f"timing {dtype} N={n} 1080p: {dt*1e3:.2f} ms/step  {n/dt:.1f} img/s  ({n/dt*328.46e9/1e12:.1f} TFLOP/s)")
