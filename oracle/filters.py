"""Oracle (TEST INFRASTRUCTURE): CPU fp32 restatements of the filtering / colour-space side of the V-channel and guided curve
models, in the reference's own operation order.

  box_filter, guided_filter, fast_guided_filter   src/mon/vision/filtering/box_filter.py:24-110, guided_filter.py:99-204; box_filter and the GuidedFilter class
                                                   (guided_filter_module: its window is `radius` pixels WIDE) are PINNED by tests/golden/guided_filter_seed161.npz
  zero_dce_v_forward                               src/mon/vision/enhance/llie/zero_dce/zero_dce_v.py:172-230
  gcenet_forward                                   src/mon/vision/enhance/llie/gcenet/gcenet.py:203-309, 460-502
  boundary_aware_prior, brightness_attention_map   src/mon/nn/modules/prior/image.py:187-331

parity unpinned: the mon package does not import here and these models have no mon_extra twin.  kornia
(``rgb_to_hsv``, ``hsv_to_rgb``, ``rgb_to_grayscale``, ``median_blur``, ``sobel``; unpinned in env/linux_console.yaml) is absent:
its published algorithms are restated below (kornia/color/hsv.py, kornia/filters/median.py, kornia/filters/sobel.py).
"""
from __future__ import annotations

import math

import torch
import torch.nn.functional as F

from .generators import _c


# ---- kornia restatements ---------------------------------------------------------------------------------------------------
def rgb_to_hsv(image: torch.Tensor, eps: float = 1e-8) -> torch.Tensor:
    max_rgb, argmax_rgb = image.max(-3)
    min_rgb, _ = image.min(-3)
    deltac = max_rgb - min_rgb
    v = max_rgb
    s = deltac / (max_rgb + eps)
    deltac = torch.where(deltac == 0, torch.ones_like(deltac), deltac)
    rc, gc, bc = torch.unbind(max_rgb.unsqueeze(-3) - image, dim=-3)
    h1 = bc - gc
    h2 = (rc - bc) + 2.0 * deltac
    h3 = (gc - rc) + 4.0 * deltac
    h = torch.stack((h1, h2, h3), dim=-3) / deltac.unsqueeze(-3)
    h = torch.gather(h, dim=-3, index=argmax_rgb.unsqueeze(-3)).squeeze(-3)
    h = (h / 6.0) % 1.0
    h = 2.0 * math.pi * h
    return torch.stack((h, s, v), dim=-3)


def hsv_to_rgb(image: torch.Tensor) -> torch.Tensor:
    h = image[..., 0, :, :] / (2 * math.pi)
    s = image[..., 1, :, :]
    v = image[..., 2, :, :]
    hi = torch.floor(h * 6) % 6
    f = ((h * 6) % 6) - hi
    one = torch.tensor(1.0)
    p = v * (one - s)
    q = v * (one - f * s)
    t = v * (one - (one - f) * s)
    hi = hi.long()
    indices = torch.stack([hi, hi + 6, hi + 12], dim=-3)
    out = torch.stack((v, q, p, p, t, v, t, v, v, q, p, p, p, p, t, v, v, q), dim=-3)
    return torch.gather(out, -3, indices)


def rgb_to_grayscale(image: torch.Tensor) -> torch.Tensor:
    r, g, b = image[:, 0:1], image[:, 1:2], image[:, 2:3]
    return 0.299 * r + 0.587 * g + 0.114 * b


def median_blur(x: torch.Tensor, k: int) -> torch.Tensor:
    """kornia.filters.median_blur: unfold with ZERO padding, median over the k*k taps (odd count: the true median)."""
    n, c, h, w = x.shape
    cols = F.unfold(x.reshape(n * c, 1, h, w), k, padding=k // 2)          # (n*c, k*k, h*w)
    return cols.median(dim=1)[0].reshape(n, c, h, w)


def sobel(x: torch.Tensor, eps: float = 1e-6) -> torch.Tensor:
    """kornia.filters.sobel(normalized=False): sqrt(gx^2 + gy^2 + eps), 3x3 Sobel kernels on the replicate-padded image."""
    kx = torch.tensor([[-1.0, 0.0, 1.0], [-2.0, 0.0, 2.0], [-1.0, 0.0, 1.0]])
    ky = kx.t()
    n, c, h, w = x.shape
    xp = F.pad(x.reshape(n * c, 1, h, w), (1, 1, 1, 1), mode="replicate")
    gx = F.conv2d(xp, kx[None, None])
    gy = F.conv2d(xp, ky[None, None])
    return torch.sqrt(gx * gx + gy * gy + eps).reshape(n, c, h, w)


# ---- box / guided filters (the reference's cumsum form) ---------------------------------------------------------------------
def _diff_x(image, r):
    left = image[:, :, r:2 * r + 1]
    middle = image[:, :, 2 * r + 1:] - image[:, :, :-2 * r - 1]
    right = image[:, :, -1:] - image[:, :, -2 * r - 1:-r - 1]
    return torch.cat([left, middle, right], dim=2)


def _diff_y(image, r):
    left = image[:, :, :, r:2 * r + 1]
    middle = image[:, :, :, 2 * r + 1:] - image[:, :, :, :-2 * r - 1]
    right = image[:, :, :, -1:] - image[:, :, :, -2 * r - 1:-r - 1]
    return torch.cat([left, middle, right], dim=3)


def box_filter(x: torch.Tensor, r: int) -> torch.Tensor:
    """box_filter.py:71-110 (sum over the (2r + 1)^2 window by two cumulative sums; r = 0 runs the same differences of cumulative sums and returns the tensor up to
    their rounding).  PINNED by tests/golden/guided_filter_seed161.npz."""
    return _diff_y(_diff_x(x.cumsum(dim=2), r).cumsum(dim=3), r)


def _gf_coeffs(x, y, r, eps):
    n = box_filter(torch.ones((1, 1, x.shape[2], x.shape[3]), dtype=x.dtype), r)
    mean_x = box_filter(x, r) / n
    mean_y = box_filter(y, r) / n
    cov_xy = box_filter(x * y, r) / n - mean_x * mean_y
    var_x = box_filter(x * x, r) / n - mean_x * mean_x
    a = cov_xy / (var_x + eps)
    return a, mean_y - a * mean_x, n


def guided_filter(x: torch.Tensor, y: torch.Tensor, r: int, eps: float = 1e-8) -> torch.Tensor:
    a, b, n = _gf_coeffs(x, y, r, eps)
    return box_filter(a, r) / n * x + box_filter(b, r) / n


def guided_filter_module(x: torch.Tensor, y: torch.Tensor, radius: int, eps: float = 1e-8) -> torch.Tensor:
    """``GuidedFilter(radius, eps)(image=x, guide=y)`` AS THE REFERENCE CLASS BEHAVES (guided_filter.py:111-146): its constructor passes ``radius`` as BoxFilter's first
    positional argument, which is ``kernel_size`` (box_filter.py:163-172), so the windows are ``radius`` pixels wide -- box radius int((radius - 1) / 2): GCENet's
    ``gf_radius = 3`` (gcenet.py:355, 391) filters with 3x3 windows, and radius 1 or 2 degenerates to the guide itself (window 1: cov = 0, A = 0, b = y).
    FastGuidedFilter passes ``radius=`` by keyword and gets what it asks for.  PINNED by tests/golden/guided_filter_seed161.npz (the class executed out of the file)."""
    return guided_filter(x, y, int((radius - 1) / 2), eps)


def fast_guided_filter(x_lr, y_lr, x_hr, r: int, eps: float = 1e-8) -> torch.Tensor:
    a, b, _ = _gf_coeffs(x_lr, y_lr, r, eps)
    size = x_hr.shape[2:]
    return F.interpolate(a, size, mode="bicubic", align_corners=True) * x_hr + F.interpolate(b, size, mode="bicubic", align_corners=True)


# ---- priors -----------------------------------------------------------------------------------------------------------------
def boundary_aware_prior(x: torch.Tensor, eps: float = 0.05) -> torch.Tensor:
    g = sobel(x, 1e-6)
    return (g / torch.max(g) > eps).float()


def brightness_attention_map(image: torch.Tensor, gamma: float = 2.5, ksize: int | None = None) -> torch.Tensor:
    if ksize:
        image = median_blur(image, ksize)
    v = rgb_to_hsv(image)[:, 2:3]
    return torch.pow(1 - v, gamma)


# ---- models -----------------------------------------------------------------------------------------------------------------
def zero_dce_v_forward(sd: dict, image: torch.Tensor, num_iters: int = 15, down_size: int = 256) -> dict:
    hsv = rgb_to_hsv(image)
    v = hsv[:, 2:3]
    v_lr = F.interpolate(v, size=(down_size, down_size))
    x1 = F.relu(_c(sd, "e_conv1", v_lr)); x2 = F.relu(_c(sd, "e_conv2", x1)); x3 = F.relu(_c(sd, "e_conv3", x2)); x4 = F.relu(_c(sd, "e_conv4", x3))
    x5 = F.relu(_c(sd, "e_conv5", torch.cat([x3, x4], 1)))
    x6 = F.relu(_c(sd, "e_conv6", torch.cat([x2, x5], 1)))
    xr = torch.tanh(_c(sd, "e_conv7", torch.cat([x1, x6], 1)))
    y = v_lr
    for r in torch.split(xr, 1, dim=1)[:num_iters]:
        y = y + r * (torch.pow(y, 2) - y)
    v_fixed = torch.clip(fast_guided_filter(v_lr, y, v, 1), 0, 1)
    hsv2 = hsv.clone()
    hsv2[:, -1:, :, :] = v_fixed
    rgb = hsv_to_rgb(hsv2)
    return {"image_v": v, "image_v_lr": v_lr, "adjust": xr, "image_v_fixed_lr": y, "image_v_fixed": v_fixed, "enhanced": rgb / torch.max(rgb)}


def gcenet_forward(sd: dict, image: torch.Tensor, depth: torch.Tensor | None, num_iters: int = 15, dba_eps: float = 0.05, gf_radius: int = 3,
                   gf_eps: float = 1e-4, bam_gamma: float = 2.6, bam_ksize: int = 9, use_depth: bool = True, use_edge: bool = True) -> dict:
    def block(k, t, act="lrelu"):
        n = f"en.e_conv{k}.conv"
        t = _c(sd, f"{n}.dw_conv", t, groups=t.shape[1])
        t = _c(sd, f"{n}.pw_conv", t, padding=0)
        return torch.tanh(t) if act == "tanh" else F.leaky_relu(t, 0.2)
    x = image
    gray = rgb_to_grayscale(image)
    edge = None
    if depth is not None and depth.shape[1] == 3:
        depth = rgb_to_grayscale(depth)
    if use_depth:
        x = torch.cat([x, depth], 1)
    if use_edge:
        edge = boundary_aware_prior(depth if depth is not None else gray, dba_eps)
        x = torch.cat([x, edge], 1)
    x1 = block(1, x); x2 = block(2, x1); x3 = block(3, x2); x4 = block(4, x3)
    x5 = block(5, torch.cat([x3, x4], 1)); x6 = block(6, torch.cat([x2, x5], 1))
    adjust = block(7, torch.cat([x1, x6], 1), "tanh")
    enhanced = image
    bam = None
    if bam_gamma in (None, 0.0):
        for _ in range(num_iters):
            enhanced = enhanced + adjust * (torch.pow(enhanced, 2) - enhanced)
    else:
        bam = brightness_attention_map(image, bam_gamma, bam_ksize)
        for _ in range(num_iters):
            bright = enhanced * (1 - bam)
            dark = enhanced * bam
            enhanced = bright + dark + adjust * (torch.pow(dark, 2) - dark)
    enhanced = guided_filter_module(image, enhanced, gf_radius, gf_eps)
    return {"adjust": adjust, "edge": edge, "bam": bam, "enhanced": enhanced}
