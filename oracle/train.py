"""Oracle (TEST INFRASTRUCTURE): torch-CPU restatement of the supervised training step.

  l1_loss / ssim / ssim_loss   src/mon/nn/loss/base.py:164-178, src/mon/nn/metric/image.py:90-237, nn/loss/image.py:834-866
  unet_forward_train           src/mon/vision/segment/unet/unet.py:35-197 under model.train() (BatchNorm on batch statistics)
  training_step                forward_loss (vision/enhance/base.py:47-57) -> torch.autograd -> clip_grad_norm_
                               (Lightning gradient_clip_val, norm) -> torch.optim.Adam (nn/model.py:498-615)

Run it in float64 for gradient parity: torch's fp32 CPU backward of this network is itself ~5e-3 (relative) away from the
float64 result (fp32 partial sums in BatchNorm's backward), far more than the HIP path's error.

ssim / ssim_loss / gauss_1d are PINNED (round 4) by tests/golden/ssim_*.npz: metric/image.py imports torchmetrics and mon.globals at module
level, so tests/golden/make_golden.py takes custom_ssim / _custom_ssim / _gaussian_filter / _fspecial_gauss_1d out of its syntax tree by name and
executes them unmodified.  The four zero-reference terms below are pinned the same way by tests/golden/zdce_losses_*.npz against L_TV / L_spa /
L_color / L_exp of the original repository's myloss.py (src/mon_extra/vision/enhance/llie/zero_dce/myloss.py; mon's TotalVariationLoss, which is what
this file restates, additionally divides by the channel count) AND by tests/golden/zdce_loss_mon_*.npz against mon's own classes: Loss of zero_dce.py:32-79
over ColorConstancyLoss / ExposureControlLoss / SpatialConsistencyLoss / TotalVariationLoss of nn/loss/image.py, executed out of their files (terms and composite).
parity unpinned: the training harness itself (lightning is not installed); its leaf arithmetic is torch's own (F.batch_norm, autograd,
torch.optim.Adam, clip_grad_norm_).
"""
from __future__ import annotations

import torch
import torch.nn.functional as F


def gauss_1d(size: int = 11, sigma: float = 1.5) -> torch.Tensor:
    coords = torch.arange(size, dtype=torch.float) - size // 2
    g = torch.exp(-(coords ** 2) / (2 * sigma ** 2))
    return g / g.sum()


def ssim(x, y, data_range=255.0, k=(0.01, 0.03), sigma=1.5, non_negative=False):
    """custom_ssim(size_average=True): mean over (n, c) of the mean SSIM map."""
    c = x.shape[1]
    win = gauss_1d(11, sigma).to(x.dtype)
    wh = win.view(1, 1, 11, 1).repeat(c, 1, 1, 1)
    ww = win.view(1, 1, 1, 11).repeat(c, 1, 1, 1)

    def filt(t):
        return F.conv2d(F.conv2d(t, wh, groups=c), ww, groups=c)
    c1, c2 = (k[0] * data_range) ** 2, (k[1] * data_range) ** 2
    mu1, mu2 = filt(x), filt(y)
    s1, s2, s12 = filt(x * x) - mu1 ** 2, filt(y * y) - mu2 ** 2, filt(x * y) - mu1 * mu2
    cs = (2 * s12 + c2) / (s1 + s2 + c2)
    m = ((2 * mu1 * mu2 + c1) / (mu1 ** 2 + mu2 ** 2 + c1)) * cs
    per = m.flatten(2).mean(-1)
    if non_negative:
        per = torch.relu(per)
    return per.mean()


def l1_loss(x, y, loss_weight=1.0):
    return loss_weight * F.l1_loss(x, y)


def ssim_loss(x, y, data_range=255.0, non_negative=False):
    return 1.0 - ssim(x, y, data_range=data_range, non_negative=non_negative)


def l1_ssim_loss(x, y, l1_weight=1.0, ssim_weight=1.0, data_range=1.0, non_negative=False):
    return l1_weight * l1_loss(x, y) + ssim_weight * ssim_loss(x, y, data_range, non_negative)


class _StorageRound(torch.autograd.Function):
    """y = x rounded to a 16-bit storage format, and the gradient arriving at y rounded the same way: what a tensor (and its
    gradient buffer) kept in bf16 / fp16 between two kernels of the engine's 16-bit training graph goes through."""

    @staticmethod
    def forward(ctx, x, dt):
        ctx.dt = dt
        return x.to(dt).to(x.dtype)

    @staticmethod
    def backward(ctx, g):
        return g.to(ctx.dt).to(g.dtype), None


def storage_rounding(dt):
    """(activation rounding, weight rounding) callables for ``unet_forward_train(..., q=...)``: activations and their gradients are
    stored in ``dt``; weights are READ in ``dt`` (the packed copies of the fp32 master weights) while their gradient stays exact."""
    act = lambda t: _StorageRound.apply(t, dt)
    wgt = lambda w: w + (w.detach().to(dt).to(w.dtype) - w.detach())
    return act, wgt


def unet_forward_train(p: dict, buffers: dict, x, upsampler="bilinear", enhance=True, q=None):
    """p: parameters (requires_grad), buffers: running statistics (updated in place, momentum 0.1).
    ``q = storage_rounding(dtype)``: restates the engine's 16-bit training graph -- every tensor between two kernels (conv output,
    norm + ReLU output, resized map) and its gradient rounded to ``dtype``, weights read in ``dtype``, arithmetic in x's dtype --
    so that a comparison isolates kernel error from storage rounding."""
    qa, qw = q if q is not None else ((lambda t: t), (lambda w: w))

    def bn(n, t):
        return F.batch_norm(t, buffers[f"{n}.running_mean"], buffers[f"{n}.running_var"], p[f"{n}.weight"], p[f"{n}.bias"], True, 0.1, 1e-5)

    def dc(n, t):
        t = qa(F.relu(bn(f"{n}.norm1", qa(F.conv2d(t, qw(p[f"{n}.conv1.weight"]), None, padding=1)))))
        return qa(F.relu(bn(f"{n}.norm2", qa(F.conv2d(t, qw(p[f"{n}.conv2.weight"]), None, padding=1)))))
    if q is not None:
        x = qa(x.detach().clone())
    x1 = dc("inc", x)
    x2 = dc("down1.conv", F.max_pool2d(x1, 2))
    x3 = dc("down2.conv", F.max_pool2d(x2, 2))
    x4 = dc("down3.conv", F.max_pool2d(x3, 2))
    x5 = dc("down4.conv", F.max_pool2d(x4, 2))

    def up(k, a, skip):
        if upsampler == "bilinear":
            a = qa(F.interpolate(a, scale_factor=2, mode="bilinear", align_corners=True))
        elif upsampler == "convtranspose":      # the reference's bilinear=False branch (unet.py:85-86): nn.ConvTranspose2d(in, in // 2, kernel_size=2, stride=2)
            a = qa(F.conv_transpose2d(a, qw(p[f"up{k}.up.weight"]), p.get(f"up{k}.up.bias"), stride=2))
        else:
            a = F.pixel_shuffle(qa(F.conv2d(a, qw(p[f"up{k}.up.0.weight"]), p[f"up{k}.up.0.bias"], padding=1)), 2)
        dy, dx = skip.shape[2] - a.shape[2], skip.shape[3] - a.shape[3]
        a = F.pad(a, [dx // 2, dx - dx // 2, dy // 2, dy - dy // 2])
        return dc(f"up{k}.conv", torch.cat([skip, a], 1))
    y = up(1, x5, x4); y = up(2, y, x3); y = up(3, y, x2); y = up(4, y, x1)
    y = qa(F.conv2d(y, qw(p["outc.conv.weight"]), p["outc.conv.bias"]))
    return torch.clamp(y, 0, 1) if enhance else y


def training_steps(state_dict: dict, batches, forward_train, loss_fn, lr=1e-4, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.0,
                   clip=0.1, world_batches=None):
    """Run len(batches) optimiser steps.  Returns per step: loss, gradients (before clipping), grad norm; and the final state.
    ``world_batches``: optional list (per step) of lists of (image, target) shards -> DDP semantics (mean of shard gradients)."""
    params = {k: v.clone().requires_grad_(True) for k, v in state_dict.items() if "running_" not in k}
    buffers = {k: v.clone() for k, v in state_dict.items() if "running_" in k}
    opt = torch.optim.Adam(list(params.values()), lr=lr, betas=betas, eps=eps, weight_decay=weight_decay)
    log = []
    for step, (image, target) in enumerate(batches):
        opt.zero_grad()
        shards = world_batches[step] if world_batches else [(image, target)]
        losses = []
        for im, tg in shards:
            pred = forward_train(params, buffers, im)
            loss = loss_fn(pred, tg) / len(shards)
            loss.backward()
            losses.append(float(loss.detach()) * len(shards))
        grads = {k: v.grad.clone() for k, v in params.items()}
        norm = float(torch.nn.utils.clip_grad_norm_(list(params.values()), clip)) if clip else None
        opt.step()
        log.append({"loss": losses, "grads": grads, "grad_norm": norm})
    final = {k: v.detach().clone() for k, v in params.items()}
    final.update({k: v.clone() for k, v in buffers.items()})
    return log, final


# ---- Zero-DCE zero-reference training (zero_dce.py:34-79, 160-191) ------------------------------------------------------------
def spatial_consistency_loss(input, target, patch: int = 4):
    """nn/loss/image.py:905-1270 with num_regions = 4: contrasts of the pooled channel means against the 4 neighbours
    (conv2d with the +1/-1 kernels, zero padding 1), squared difference between the two images, mean."""
    org, enh = F.avg_pool2d(input.mean(1, keepdim=True), patch), F.avg_pool2d(target.mean(1, keepdim=True), patch)
    k = torch.zeros(4, 1, 3, 3, dtype=input.dtype)
    k[:, 0, 1, 1] = 1
    k[0, 0, 1, 0] = -1; k[1, 0, 1, 2] = -1; k[2, 0, 0, 1] = -1; k[3, 0, 2, 1] = -1          # left, right, up, down
    d = F.conv2d(org, k, padding=1) - F.conv2d(enh, k, padding=1)
    return (d ** 2).sum(1, keepdim=True).mean()


def exposure_control_loss(input, patch: int = 16, mean_val: float = 0.6):
    """nn/loss/image.py:534-568."""
    return ((F.avg_pool2d(input.mean(1, keepdim=True), patch) - mean_val) ** 2).mean()


def color_constancy_loss(input):
    """nn/loss/image.py:191-217 (d_* are already squared differences, squared again under the root)."""
    m = input.mean((2, 3), keepdim=True)
    mr, mg, mb = torch.split(m, 1, dim=1)
    drg, drb, dgb = (mr - mg) ** 2, (mr - mb) ** 2, (mb - mg) ** 2
    return torch.pow(drg ** 2 + drb ** 2 + dgb ** 2, 0.5).mean()


def total_variation_loss(x):
    """nn/loss/image.py:1355-1392."""
    b, c, h, w = x.shape
    count_h, count_w = c * (h - 1) * w, c * h * (w - 1)
    h_tv = ((x[:, :, 1:, :] - x[:, :, :h - 1, :]) ** 2).sum()
    w_tv = ((x[:, :, :, 1:] - x[:, :, :, :w - 1]) ** 2).sum()
    return 2 * (h_tv / count_h + w_tv / count_w) / b


def zero_dce_loss(input, adjust, enhance, spa_weight=1.0, exp_weight=10.0, col_weight=5.0, tva_weight=200.0, exp_mean_val=0.6):
    """zero_dce.py:65-79: spa(input=enhance, target=input) + exp(enhance) + col(enhance) + tva(adjust)."""
    return (spa_weight * spatial_consistency_loss(enhance, input) + exp_weight * exposure_control_loss(enhance, 16, exp_mean_val)
            + col_weight * color_constancy_loss(enhance) + tva_weight * total_variation_loss(adjust))


def zero_dce_forward_train(p: dict, buffers: dict, x, num_iters: int = 8):
    """zero_dce.py:173-191; returns (adjust, enhanced)."""
    c = lambda k, t: F.conv2d(t, p[f"e_conv{k}.weight"], p[f"e_conv{k}.bias"], padding=1)
    x1 = F.relu(c(1, x)); x2 = F.relu(c(2, x1)); x3 = F.relu(c(3, x2)); x4 = F.relu(c(4, x3))
    x5 = F.relu(c(5, torch.cat([x3, x4], 1)))
    x6 = F.relu(c(6, torch.cat([x2, x5], 1)))
    xr = torch.tanh(c(7, torch.cat([x1, x6], 1)))
    y = x
    for i in range(num_iters):
        r = xr[:, 3 * i:3 * i + 3]
        y = y + r * (torch.pow(y, 2) - y)
    return xr, y
