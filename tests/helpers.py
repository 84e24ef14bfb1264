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
