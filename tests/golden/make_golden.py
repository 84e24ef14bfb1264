#!/usr/bin/env python
"""Generate tests/golden/*.npz from the REFERENCE's own model files.

Runs ONLY in the build container (needs /root/reference).  It importlib-loads
the pure-torch reference files
    src/mon_extra/vision/enhance/llie/zero_dce/model.py      (enhance_net_nopool)
    src/mon_extra/vision/enhance/llie/zero_dce++/model.py    (enhance_net_nopool, CSDN_Tem)
    src/mon_extra/vision/enhance/llie/llunet++/model.py      (NestedUNet)
loads seeded weights into them, runs their forward on CPU fp32 and stores
inputs / weights / outputs as small fixtures.  The fixtures are data only; no
reference source is copied.  Usage:  python tests/golden/make_golden.py
"""
from __future__ import annotations

import importlib.util
import os
import sys

import numpy as np
import torch

REF = os.environ.get("IE_REFERENCE_ROOT", "/root/reference")
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))

from oracle import zero_dce as o_zdce  # noqa: E402  (seeded weight rule lives in the oracle)


def _load(rel, name):
    spec = importlib.util.spec_from_file_location(name, os.path.join(REF, rel))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def _sample_crop(h, w, y0=200, x0=250):
    """Crop of the reference's sample low-light image project/enhance/data/10.png."""
    from PIL import Image
    im = np.asarray(Image.open(os.path.join(REF, "project/enhance/data/10.png")).convert("RGB"))
    crop = im[y0:y0 + h, x0:x0 + w].astype(np.float32) / 255.0  # to_image_tensor(normalize): /255
    return torch.from_numpy(crop).permute(2, 0, 1)[None].contiguous()


def _u8(x):
    return x.mul(255).add(0.5).clamp(0, 255).to(torch.uint8).permute(0, 2, 3, 1).contiguous().numpy()


def gen_zero_dce():
    mod = _load("src/mon_extra/vision/enhance/llie/zero_dce/model.py", "ref_zero_dce")
    torch.manual_seed(0)
    cases = {
        # name: (state_dict, input)
        "init_rand_1x64x96": (o_zdce.init_state_dict(seed=1), torch.rand(1, 3, 64, 96)),
        "trained_lowlight_2x40x72": (o_zdce.init_state_dict(seed=2, gain=4.0),
                                     torch.rand(2, 3, 40, 72) ** 2.2 * 0.25),
        "trained_sample_1x48x80": (o_zdce.init_state_dict(seed=3, gain=4.0), _sample_crop(48, 80)),
        # ragged: W not a multiple of any tile width, H odd
        "trained_rand_1x37x131": (o_zdce.init_state_dict(seed=4, gain=3.0), torch.rand(1, 3, 37, 131)),
    }
    for name, (sd, x) in cases.items():
        net = mod.enhance_net_nopool()
        net.load_state_dict(sd, strict=True)
        net.eval()
        with torch.no_grad():
            e1, e, r = net(x)
        out = {f"w.{k}": v.numpy() for k, v in sd.items()}
        out.update(image=x.numpy(), enhance_image_1=e1.numpy(), enhanced=e.numpy(), adjust=r.numpy(),
                   enhanced_u8=_u8(e))
        path = os.path.join(HERE, f"zero_dce_{name}.npz")
        np.savez_compressed(path, **out)
        print("wrote", path, os.path.getsize(path) // 1024, "KiB",
              "| |r|max", float(r.abs().max()), "enh range", float(e.min()), float(e.max()))


from tests.helpers import seeded_state_dict as _seeded  # noqa: E402


def gen_zero_dcepp():
    mod = _load("src/mon_extra/vision/enhance/llie/zero_dce++/model.py", "ref_zero_dcepp")
    for name, scale, x, gain in (("s1_1x40x56", 1, torch.rand(1, 3, 40, 56, generator=torch.Generator().manual_seed(5)), 8.0),
                                 ("s2_2x48x64", 2, torch.rand(2, 3, 48, 64, generator=torch.Generator().manual_seed(6)) ** 2.2 * 0.3, 8.0)):
        net = mod.enhance_net_nopool(scale)
        shapes = {k: tuple(v.shape) for k, v in net.state_dict().items()}
        sd = _seeded(shapes, seed=21 + scale, std=0.02 * gain)
        net.load_state_dict(sd, strict=True)
        net.eval()
        with torch.no_grad():
            e, r = net(x)
        out = {f"w.{k}": v.numpy() for k, v in sd.items()}       # original key names (depth_conv / point_conv)
        out.update(image=x.numpy(), enhanced=e.numpy(), adjust=r.numpy(), scale_factor=np.array(scale))
        path = os.path.join(HERE, f"zero_dcepp_{name}.npz")
        np.savez_compressed(path, **out)
        print("wrote", path, os.path.getsize(path) // 1024, "KiB | |r|max", float(r.abs().max()))


def gen_llunetpp():
    """LLUnet++ has 29 M parameters: the weights are NOT stored; they are regenerated from the seed by
    oracle-side code (tests/helpers.llunetpp_seeded_weights) and only input/output travel."""
    mod = _load("src/mon_extra/vision/enhance/llie/llunet++/model.py", "ref_llunetpp")
    net = mod.NestedUNet()
    shapes = {k: tuple(v.shape) for k, v in net.state_dict().items()}
    sd = _seeded(shapes, seed=31)
    net.load_state_dict(sd, strict=True)
    net.eval()
    x = torch.rand(1, 3, 32, 48, generator=torch.Generator().manual_seed(7))
    with torch.no_grad():
        y = net(x)
    path = os.path.join(HERE, "llunetpp_seed31_1x32x48.npz")
    np.savez_compressed(path, image=x.numpy(), enhanced=y.numpy(), seed=np.array(31),
                        key_order=np.array(list(shapes.keys())), key_shapes=np.array([str(s) for s in shapes.values()]))
    print("wrote", path, os.path.getsize(path) // 1024, "KiB | out range", float(y.min()), float(y.max()), "mean", float(y.mean()))


if __name__ == "__main__":
    torch.set_num_threads(8)
    which = sys.argv[1:] or ["zero_dce", "zero_dcepp", "llunetpp"]
    if "zero_dce" in which:
        gen_zero_dce()
    if "zero_dcepp" in which:
        gen_zero_dcepp()
    if "llunetpp" in which:
        gen_llunetpp()
