"""Losses of the supervised training path, computed by the library's kernels.

  l1_loss     L1Loss     src/mon/nn/loss/base.py:164-178          loss_weight * mean|input - target|
  ssim_loss   SSIMLoss   src/mon/nn/loss/image.py:834-866         1 - custom_ssim(...)  (loss_weight is NOT applied, as in
                                                                    the reference's forward); CustomSSIM: nn/metric/image.py:316-346
  l1_ssim_loss  L1SSIMLoss  engine composite for BASELINE config 5: l1_weight * L1 + ssim_weight * SSIMLoss
  zero_dce_loss ZeroDCELoss src/mon/vision/enhance/llie/zero_dce/zero_dce.py:34-79: spa + 10 exp + 5 col + 200 tvA (zero-reference;
                            SpatialConsistencyLoss nn/loss/image.py:905-1270, ExposureControlLoss :534-568, ColorConstancyLoss
                            :191-217, TotalVariationLoss :1355-1392); called as loss(input, adjust, enhance)

Each loss is called as ``loss(input, target)`` and returns a ``LossResult``: ``float(result)`` is the value (one device
sync); ``result.backward_into(grad)`` enqueues d(loss)/d(input) into ``grad`` (same memory layout as ``input``).
Inputs are cuda fp32 (N,C,H,W) tensors in any dense layout (NCHW or channels_last); strides are passed to the kernels.
"""
from __future__ import annotations

import torch

from . import capi
from .registry import Factory

LOSSES = Factory("Losses")


def _same_layout(input: torch.Tensor, target: torch.Tensor) -> torch.Tensor:
    if input.shape != target.shape:
        raise ValueError(f"`image1` and `image2` must have the same dimensions, but got {tuple(input.shape)} and {tuple(target.shape)}.")
    if input.dim() != 4:
        raise ValueError(f"`image1` and `image2` must be 4D tensors, but got {tuple(input.shape)}.")
    if not input.is_cuda or input.dtype != torch.float32:
        raise capi.IEError("losses run on the HIP device on float32 tensors (no CPU path)")
    target = target.to(input.device, torch.float32)
    if target.stride() != input.stride():
        t = torch.empty_like(input)          # preserve_format: same strides as input
        t.copy_(target)
        target = t
    return target


class LossResult:
    def __init__(self, value_fn, backward_fn, seed_fn=None):
        self._value_fn, self._backward_fn, self._value = value_fn, backward_fn, None
        self.seed_fn = seed_fn        # multi-input losses: seed_fn(tape, scale) puts the gradients of their inputs on the tape

    def __float__(self):
        if self._value is None:
            self._value = float(self._value_fn())
        return self._value

    def item(self):
        return float(self)

    def backward_into(self, grad: torch.Tensor, accumulate: bool = False, scale: float = 1.0):
        self._backward_fn(grad, accumulate, scale)


class Loss:
    reductions = ["none", "mean", "sum"]

    def __init__(self, loss_weight: float = 1.0, reduction: str = "mean"):
        if reduction not in self.reductions:
            raise ValueError(f"`reduction` must be one of: {self.reductions}, but got {reduction}.")
        self.loss_weight, self.reduction = loss_weight, reduction

    def __call__(self, input, target):
        return self.forward(input, target)


@LOSSES.register(name="l1_loss")
class L1Loss(Loss):
    def forward(self, input: torch.Tensor, target: torch.Tensor) -> LossResult:
        if self.reduction == "none":
            raise capi.IEError("l1_loss: reduction='none' is not implemented on the device path")
        target = _same_layout(input, target)
        n = input.numel()
        acc = torch.empty(1, device=input.device, dtype=torch.float64)
        div = float(n) if self.reduction == "mean" else 1.0
        L = capi.lib()
        with torch.cuda.device(input.device):
            capi.check(L.ie_l1_loss_f32(capi.ptr(input), capi.ptr(target), n, capi.ptr(acc), None, 0.0, 0, capi.current_stream()), "ie_l1_loss_f32")

        def backward(grad, accumulate, scale):
            tmp = torch.empty(1, device=input.device, dtype=torch.float64)
            with torch.cuda.device(input.device):
                capi.check(L.ie_l1_loss_f32(capi.ptr(input), capi.ptr(target), n, capi.ptr(tmp), capi.ptr(grad),
                                            scale * self.loss_weight / div, int(accumulate), capi.current_stream()), "ie_l1_loss_f32")
        return LossResult(lambda: self.loss_weight * acc.item() / div, backward)


@LOSSES.register(name="ssim_loss")
class SSIMLoss(Loss):
    def __init__(self, data_range: float = 255, size_average: bool = True, window_size: int = 11, window_sigma: float = 1.5,
                 channel: int = 3, spatial_dims: int = 2, k=(0.01, 0.03), non_negative_ssim: bool = False, loss_weight: float = 1.0,
                 reduction: str = "mean"):
        super().__init__(loss_weight, reduction)
        if window_size != 11:
            raise capi.IEError("ssim_loss: the kernel implements the reference's 11-tap window only")
        if not size_average:
            raise capi.IEError("ssim_loss: size_average=False is not implemented on the device path")
        self.data_range, self.sigma, self.k, self.non_negative = float(data_range), float(window_sigma), tuple(k), bool(non_negative_ssim)

    def forward(self, input: torch.Tensor, target: torch.Tensor, keep_grad_state: bool = True) -> LossResult:
        target = _same_layout(input, target)
        n, c, h, w = input.shape
        sn, sc, sh, sw = input.stride()
        cnt = (h - 10) * (w - 10)
        sums = torch.empty(n * c, device=input.device, dtype=torch.float64)
        dws = torch.empty(3 * n * c * max(cnt, 0), device=input.device, dtype=torch.float32) if keep_grad_state else None
        L = capi.lib()
        with torch.cuda.device(input.device):
            capi.check(L.ie_ssim_forward_f32(capi.ptr(input), capi.ptr(target), n, c, h, w, sn, sc, sh, sw, self.data_range, self.k[0],
                                             self.k[1], self.sigma, capi.ptr(sums), capi.ptr(dws), capi.current_stream()), "ie_ssim_forward_f32")

        def value():
            per = sums.cpu() / cnt                      # ssim_per_channel (N*C)
            if self.non_negative:
                per = per.clamp_min(0.0)
            return 1.0 - per.mean().item()

        def backward(grad, accumulate, scale):
            if dws is None:
                raise capi.IEError("ssim_loss: forward was run with keep_grad_state=False")
            with torch.cuda.device(input.device):
                capi.check(L.ie_ssim_backward_f32(capi.ptr(input), capi.ptr(target), capi.ptr(dws), capi.ptr(sums), capi.ptr(grad), n, c,
                                                  h, w, sn, sc, sh, sw, self.sigma, -scale / (n * c * cnt), int(self.non_negative),
                                                  int(accumulate), capi.current_stream()), "ie_ssim_backward_f32")
        return LossResult(value, backward)


@LOSSES.register(name="l1_ssim_loss")
class L1SSIMLoss(Loss):
    """BASELINE.json configs[4]: "U-Net-64 training with L1+SSIM loss"."""

    def __init__(self, l1_weight: float = 1.0, ssim_weight: float = 1.0, data_range: float = 1.0, non_negative_ssim: bool = False,
                 loss_weight: float = 1.0, reduction: str = "mean"):
        super().__init__(loss_weight, reduction)
        self.l1_weight, self.ssim_weight = l1_weight, ssim_weight
        self.l1 = L1Loss(reduction=reduction)
        self.ssim = SSIMLoss(data_range=data_range, non_negative_ssim=non_negative_ssim)

    def forward(self, input, target) -> LossResult:
        a, b = self.l1(input, target), self.ssim(input, target)
        self.last_terms = (a, b)

        def backward(grad, accumulate, scale):
            a.backward_into(grad, accumulate, scale * self.loss_weight * self.l1_weight)
            b.backward_into(grad, True, scale * self.loss_weight * self.ssim_weight)
        return LossResult(lambda: self.loss_weight * (self.l1_weight * float(a) + self.ssim_weight * float(b)), backward)


@LOSSES.register(name="zero_dce_loss")
class ZeroDCELoss(Loss):
    """zero_dce.py:34-79.  ``input``/``enhance``: NCHW fp32 (N,3,H,W); ``adjust``: the curve maps, logical (N,3*iters,H,W) in
    channels_last memory (NHWC), as the training graph produces them."""

    zero_reference = True
    grad_scale = 1.0          # set by the Trainer (its AMP loss scale) before the forward pass

    def __init__(self, spa_weight: float = 1.0, exp_patch_size: int = 16, exp_mean_val: float = 0.6, exp_weight: float = 10.0,
                 col_weight: float = 5.0, tva_weight: float = 200.0, reduction: str = "mean", loss_weight: float = 1.0):
        super().__init__(loss_weight, reduction)
        if exp_patch_size != 16 or reduction != "mean":
            raise capi.IEError("zero_dce_loss: the kernels implement the reference defaults (exposure patch 16, reduction 'mean')")
        self.spa_weight, self.exp_weight, self.col_weight, self.tva_weight, self.exp_mean_val = spa_weight, exp_weight, col_weight, tva_weight, exp_mean_val

    def forward(self, input: torch.Tensor, adjust: torch.Tensor, enhance: torch.Tensor) -> LossResult:
        for t in (input, enhance):
            if not t.is_cuda or t.dtype != torch.float32 or t.dim() != 4 or t.shape[1] != 3 or not t.is_contiguous():
                raise capi.IEError("zero_dce_loss: `input` and `enhance` must be contiguous cuda float32 (N,3,H,W) tensors")
        if tuple(input.shape) != tuple(enhance.shape):
            raise ValueError(f"`input` and `enhance` must have the same dimensions, but got {tuple(input.shape)} and {tuple(enhance.shape)}.")
        if not adjust.is_cuda or adjust.dtype != torch.float32 or not adjust.is_contiguous(memory_format=torch.channels_last):
            raise capi.IEError("zero_dce_loss: `adjust` must be a channels_last cuda float32 tensor")
        n, _, h, w = enhance.shape
        c = adjust.shape[1]
        L = capi.lib()
        sums = torch.empty(3 + 3 * n, device=enhance.device, dtype=torch.float64)
        tv = torch.empty(2, device=enhance.device, dtype=torch.float64)
        ws = torch.empty(L.ie_zref_workspace_floats(n, h, w), device=enhance.device, dtype=torch.float32)
        g_enh = torch.empty_like(enhance)
        g_adj = torch.empty_like(adjust, memory_format=torch.channels_last)
        k = self.loss_weight
        gs = float(self.grad_scale)           # AMP loss scale of the trainer: folded into the gradient weights, not into the value
        with torch.cuda.device(enhance.device):
            capi.check(L.ie_zref_image_losses_f32(capi.ptr(enhance), capi.ptr(input), n, h, w, self.exp_mean_val, gs * k * self.spa_weight,
                                                  gs * k * self.exp_weight, gs * k * self.col_weight, capi.ptr(sums), capi.ptr(ws), capi.ptr(g_enh), 0,
                                                  capi.current_stream()), "ie_zref_image_losses_f32")
            capi.check(L.ie_tv_loss_nhwc_f32(capi.ptr(adjust), n, h, w, c, gs * k * self.tva_weight, capi.ptr(tv), capi.ptr(g_adj), 0,
                                             capi.current_stream()), "ie_tv_loss_nhwc_f32")

        def terms():
            s, t = sums[:3].cpu(), tv.cpu()
            return {"spa": float(s[0]) / (n * (h // 4) * (w // 4)), "exp": float(s[1]) / (n * (h // 16) * (w // 16)), "col": float(s[2]) / n,
                    "tva": 2.0 * (float(t[0]) / (c * (h - 1) * w) + float(t[1]) / (c * h * (w - 1))) / n}

        def value():
            t = terms()
            self.last_terms = t
            return k * (self.spa_weight * t["spa"] + self.exp_weight * t["exp"] + self.col_weight * t["col"] + self.tva_weight * t["tva"])

        def seed(tape, scale=1.0):
            if scale != gs:
                raise capi.IEError("zero_dce_loss: the gradients were computed with grad_scale %g, not %g (set loss.grad_scale before forward)" % (gs, scale))
            tape.seed(enhance, g_enh)
            tape.seed(adjust, g_adj)

        def backward(grad, accumulate, scale):
            raise capi.IEError("zero_dce_loss has two differentiable inputs: use result.seed_fn(tape)")
        return LossResult(value, backward, seed)

    def __call__(self, input, adjust, enhance):
        return self.forward(input, adjust, enhance)
