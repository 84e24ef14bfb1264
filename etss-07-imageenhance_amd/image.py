"""uint8 <-> float image edges (src/mon/core/image/utils.py:766-847, core/image/io.py:127-145)
executed by the HIP library on device tensors."""
from __future__ import annotations

import torch

from . import capi


def to_uint8(x: torch.Tensor, rule: str = "save_image") -> torch.Tensor:
    """NCHW float [0,1] (cuda) -> NHWC uint8.  ``rule="save_image"``: torchvision's
    ``clamp(x*255+0.5, 0, 255)`` used by ``write_image`` (io.py:139-140); ``"truncate"``:
    ``to_image_nparray(denormalize=True)``'s ``(x*255).astype(uint8)`` (utils.py:795) on the
    clamped value."""
    if rule not in ("save_image", "truncate"):
        raise ValueError(f"`rule` must be 'save_image' or 'truncate', but got {rule}.")
    if not x.is_cuda:
        raise capi.IEError("to_uint8 runs on the HIP device only; move the tensor to cuda.")
    x = x.to(torch.float32).contiguous()
    if x.dim() == 3:
        x = x[None]
    n, c, h, w = x.shape
    out = torch.empty((n, h, w, c), device=x.device, dtype=torch.uint8)
    with torch.cuda.device(x.device):
        capi.check(capi.lib().ie_nchw_f32_to_u8_nhwc(capi.ptr(x), capi.ptr(out), n, c, h, w,
                                                     0 if rule == "save_image" else 1, capi.current_stream()),
                   "ie_nchw_f32_to_u8_nhwc")
    return out


def to_float(x: torch.Tensor) -> torch.Tensor:
    """NHWC uint8 (cuda) -> NCHW float32 /255 (``to_image_tensor(normalize=True)``, utils.py:801-847)."""
    if not x.is_cuda or x.dtype != torch.uint8:
        raise capi.IEError("to_float expects a cuda uint8 NHWC tensor.")
    x = x.contiguous()
    if x.dim() == 3:
        x = x[None]
    n, h, w, c = x.shape
    out = torch.empty((n, c, h, w), device=x.device, dtype=torch.float32)
    with torch.cuda.device(x.device):
        capi.check(capi.lib().ie_u8_nhwc_to_nchw_f32(capi.ptr(x), capi.ptr(out), n, c, h, w, capi.current_stream()),
                   "ie_u8_nhwc_to_nchw_f32")
    return out


def psnr(a: torch.Tensor, b: torch.Tensor) -> float:
    """10*log10(1/MSE) on [0,1] floats (the build's own PSNR definition)."""
    mse = torch.mean((a.double() - b.double()) ** 2).item()
    return float("inf") if mse == 0 else 10.0 * __import__("math").log10(1.0 / mse)
