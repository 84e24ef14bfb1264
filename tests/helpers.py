"""Shared test helpers (CPU side)."""
import glob
import os

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLDEN = os.path.join(ROOT, "tests", "golden")


def load_golden(path):
    g = np.load(path)
    sd = {k[2:]: torch.from_numpy(g[k]) for k in g.files if k.startswith("w.")}
    rest = {k: g[k] for k in g.files if not k.startswith("w.")}
    return sd, rest


def zero_dce_goldens():
    return sorted(glob.glob(os.path.join(GOLDEN, "zero_dce_*.npz")))


def rel_err(a, b):
    """max |a-b| / max |b|  (the 'relative fp32' measure the parity bound is stated in)."""
    a = torch.as_tensor(a).double()
    b = torch.as_tensor(b).double()
    return float((a - b).abs().max() / b.abs().max().clamp_min(1e-30))


def psnr(a, b):
    a = torch.as_tensor(a).double()
    b = torch.as_tensor(b).double()
    mse = float(((a - b) ** 2).mean())
    return float("inf") if mse == 0 else 10.0 * np.log10(1.0 / mse)


def seeded_state_dict(shapes: dict, seed: int, std=None) -> dict:
    """Deterministic weights for architectures too large to store as fixtures (used by
    tests/golden/make_golden.py and by the tests): U(+-1/sqrt(fan_in)) conv weights (or N(0, std)),
    norm scale ~ N(1, 0.1), everything else U(+-0.1); ``final.bias`` lifted so outputs are not all clamped."""
    g = torch.Generator().manual_seed(seed)
    sd = {}
    for k, shp in shapes.items():
        shp = tuple(shp)
        if len(shp) == 4:
            bound = 1.0 / (shp[1] * shp[2] * shp[3]) ** 0.5
            sd[k] = (torch.rand(shp, generator=g) * 2 - 1) * bound if std is None else torch.randn(shp, generator=g) * std
        elif ".norm" in k and k.endswith(".weight"):
            sd[k] = 1.0 + torch.randn(shp, generator=g) * 0.1
        else:
            sd[k] = (torch.rand(shp, generator=g) * 2 - 1) * 0.1
    if "final.bias" in sd:
        sd["final.bias"] = sd["final.bias"] + 0.45
    return sd


def seeded_bn_state_dict(shapes: dict, seed: int) -> dict:
    """``seeded_state_dict`` for networks with BatchNorm: a valid, non-trivial eval-mode state (running_var in [0.5, 1.5),
    running_mean ~ N(0, 0.1), scale ~ N(1, 0.1)) for every norm layer, whatever its key spelling (``*.norm1.*`` or the
    ``nn.Sequential`` indices of the original U-Net); ``num_batches_tracked`` = 0."""
    sd = seeded_state_dict({k: s for k, s in shapes.items() if len(tuple(s)) > 0}, seed)
    g = torch.Generator().manual_seed(seed + 7919)
    for k, shp in shapes.items():
        shp = tuple(shp)
        if k.endswith("running_var"):
            sd[k] = torch.rand(shp, generator=g) + 0.5
        elif k.endswith("running_mean"):
            sd[k] = torch.randn(shp, generator=g) * 0.1
        elif k.endswith("num_batches_tracked"):
            sd[k] = torch.tensor(0)
        elif k.endswith(".weight") and len(shp) == 1 and (k[:-6] + "running_var") in shapes:
            sd[k] = 1.0 + torch.randn(shp, generator=g) * 0.1
    if "outc.conv.weight" in sd:          # U-Net head: spread the logits over [0, 1] and beyond so a clamp behind it bites on both sides
        sd["outc.conv.weight"] = sd["outc.conv.weight"] * 6.0
        sd["outc.conv.bias"] = sd["outc.conv.bias"] + 0.45
    return sd
