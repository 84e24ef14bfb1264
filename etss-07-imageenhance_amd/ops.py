"""Thin host wrappers over the fp32 NHWC building-block entry points of ``libie_hip.so``.

Tensors are torch CUDA tensors of logical shape (N, C, H, W) in ``torch.channels_last`` memory
format, i.e. physically NHWC -- the layout every kernel of the library works in.  torch is the
container (allocation, stream); each function below is one C-ABI call.  The reference equivalents are
the ``mon.nn`` aliases of ``torch.nn`` layers (src/mon/nn/modules/*); citations are in include/ie_hip.h.
"""
from __future__ import annotations

import ctypes as C

import torch

from . import capi

CL = torch.channels_last

# MAC counter used by VisionModel.compute_efficiency_score (the reference profiles with thop, which counts one multiply-accumulate
# per "FLOP"): a list with one running total, or None when no profile is being taken
_mac_counter = None


def _count_macs(n, ho, wo, cout, cin_per_group, kh, kw):
    if _mac_counter is not None:
        _mac_counter[0] += n * ho * wo * cout * cin_per_group * kh * kw


def _chk(t: torch.Tensor) -> torch.Tensor:
    if not t.is_cuda or t.dtype != torch.float32 or t.dim() != 4:
        raise capi.IEError(f"expected a cuda float32 (N,C,H,W) tensor, got {tuple(t.shape)} {t.dtype} {t.device}")
    if not t.is_contiguous(memory_format=CL):
        raise capi.IEError("expected channels_last (NHWC) memory; use ops.to_nhwc()")
    return t


def _chk_any(t: torch.Tensor) -> torch.Tensor:
    """fp32 or 16-bit channels_last tensor (the streaming ops dispatch on dtype)."""
    if not t.is_cuda or t.dtype not in (torch.float32, torch.float16, torch.bfloat16) or t.dim() != 4:
        raise capi.IEError(f"expected a cuda fp32/fp16/bf16 (N,C,H,W) tensor, got {tuple(t.shape)} {t.dtype} {t.device}")
    if not t.is_contiguous(memory_format=CL):
        raise capi.IEError("expected channels_last (NHWC) memory; use ops.to_nhwc()")
    return t


def _dt(t: torch.Tensor) -> int:
    return {torch.float32: capi.IE_F32, torch.float16: capi.IE_F16, torch.bfloat16: capi.IE_BF16}[t.dtype]


def empty_nhwc(n, c, h, w, device, dtype=torch.float32) -> torch.Tensor:
    return torch.empty((n, c, h, w), device=device, dtype=dtype, memory_format=CL)


def to_nhwc(x: torch.Tensor) -> torch.Tensor:
    """NCHW-contiguous fp32 -> channels_last (ie_nchw_f32_to_nhwc)."""
    x = x.to(torch.float32)
    if x.dim() == 3:
        x = x[None]
    n, c, h, w = x.shape
    if x.is_contiguous(memory_format=CL) and (c > 1 or not x.is_contiguous()):
        return x
    x = x.contiguous()
    y = empty_nhwc(n, c, h, w, x.device)
    with torch.cuda.device(x.device):
        capi.check(capi.lib().ie_nchw_f32_to_nhwc(capi.ptr(x), capi.ptr(y), capi.IE_F32, n, c, h, w, c, capi.current_stream()),
                   "ie_nchw_f32_to_nhwc")
    return y


def to_nchw(x: torch.Tensor) -> torch.Tensor:
    """channels_last -> NCHW-contiguous fp32 (ie_nhwc_to_nchw_f32)."""
    _chk(x)
    n, c, h, w = x.shape
    y = torch.empty((n, c, h, w), device=x.device, dtype=torch.float32)
    with torch.cuda.device(x.device):
        capi.check(capi.lib().ie_nhwc_to_nchw_f32(capi.ptr(x), capi.IE_F32, capi.ptr(y), n, c, h, w, c, capi.current_stream()),
                   "ie_nhwc_to_nchw_f32")
    return y


class ConvWeights:
    """Device-resident HWIO fp32 weights (+ bias) of one Conv2d, repacked once from the checkpoint's OIHW."""

    def __init__(self, w_oihw: torch.Tensor, bias: torch.Tensor | None, device):
        w = w_oihw.detach().to("cpu", torch.float32).contiguous()
        self.cout, self.cin_g, self.kh, self.kw = w.shape
        self.w = torch.empty(self.kh * self.kw * self.cin_g * self.cout, device=device, dtype=torch.float32)
        with torch.cuda.device(device):
            capi.check(capi.lib().ie_pack_weight_hwio_f32(capi.ptr(w), self.cout, self.cin_g, self.kh, self.kw, capi.ptr(self.w),
                                                          capi.current_stream()), "ie_pack_weight_hwio_f32")
        self.b = None if bias is None else bias.detach().to(device, torch.float32).contiguous()


def fold_bn(w: torch.Tensor, b: torch.Tensor | None, gamma, beta, mean, var, eps: float = 1e-5):
    """Eval-mode BatchNorm2d folded into the preceding conv: host-side, fp64."""
    s = (gamma.double() / torch.sqrt(var.double() + eps))
    w2 = (w.double() * s.view(-1, 1, 1, 1)).float()
    b0 = torch.zeros_like(mean, dtype=torch.float64) if b is None else b.double()
    b2 = ((b0 - mean.double()) * s + beta.double()).float()
    return w2, b2


def conv2d(srcs, cw: ConvWeights, stride=1, pad=1, pad_mode="zeros", act="none", slope=0.0, residual=None,
           clamp=None) -> torch.Tensor:
    """Conv2d over the channel concat of ``srcs`` (list of tensors or one tensor)."""
    if isinstance(srcs, torch.Tensor):
        srcs = [srcs]
    for s in srcs:
        _chk(s)
    n, _, h, w = srcs[0].shape
    cs = [int(s.shape[1]) for s in srcs]
    if sum(cs) != cw.cin_g:
        raise capi.IEError(f"conv2d: sources carry {sum(cs)} channels, weights expect {cw.cin_g}")
    d = capi.ConvNDesc(n, h, w, len(srcs), cw.cout, cw.kh, cw.kw, stride, pad, capi.PAD[pad_mode], capi.ACT[act], slope,
                       0 if clamp is None else 1, 0.0 if clamp is None else clamp[0], 0.0 if clamp is None else clamp[1], 0, 0)
    ho = (h + 2 * pad - cw.kh) // stride + 1
    wo = (w + 2 * pad - cw.kw) // stride + 1
    out = empty_nhwc(n, cw.cout, ho, wo, srcs[0].device)
    _count_macs(n, ho, wo, cw.cout, cw.cin_g, cw.kh, cw.kw)
    if residual is not None:
        _chk(residual)
        if tuple(residual.shape) != tuple(out.shape):
            raise capi.IEError("conv2d: residual shape mismatch")
    ptrs = (C.c_void_p * len(srcs))(*[s.data_ptr() for s in srcs])
    chans = (C.c_int * len(srcs))(*cs)
    with torch.cuda.device(out.device):
        capi.check(capi.lib().ie_convn_nhwc_f32(C.byref(d), ptrs, chans, capi.ptr(cw.w), capi.ptr(cw.b), capi.ptr(residual),
                                                capi.ptr(out), capi.current_stream()), "ie_convn_nhwc_f32")
    return out


def depthwise_conv2d(x, cw: ConvWeights, stride=1, pad=1, act="none") -> torch.Tensor:
    _chk(x)
    n, c, h, w = x.shape
    d = capi.Conv2dDesc(n, h, w, c, 0, c, cw.kh, cw.kw, stride, pad, 0, c, capi.ACT[act], 0.0, 0, 0)
    ho = (h + 2 * pad - cw.kh) // stride + 1
    wo = (w + 2 * pad - cw.kw) // stride + 1
    out = empty_nhwc(n, c, ho, wo, x.device)
    _count_macs(n, ho, wo, c, 1, cw.kh, cw.kw)
    with torch.cuda.device(x.device):
        capi.check(capi.lib().ie_conv2d_nhwc_f32(C.byref(d), capi.ptr(x), None, capi.ptr(cw.w), capi.ptr(cw.b), capi.ptr(out),
                                                 capi.current_stream()), "ie_conv2d_nhwc_f32(depthwise)")
    return out


def instance_norm(x, gamma=None, beta=None, eps=1e-5, act="none", slope=0.0, c_begin=0, c_count=None) -> torch.Tensor:
    _chk_any(x)
    n, c, h, w = x.shape
    c_count = c - c_begin if c_count is None else c_count
    y = torch.empty_like(x, memory_format=CL)
    ws = torch.empty(3 * n * c_count, device=x.device, dtype=torch.float64)
    with torch.cuda.device(x.device):
        capi.check(capi.lib().ie_instnorm_nhwc(capi.ptr(x), capi.ptr(gamma), capi.ptr(beta), capi.ptr(y), _dt(x), capi.ptr(ws), n, h, w,
                                               c, c_begin, c_count, eps, capi.ACT[act], slope, capi.current_stream()),
                   "ie_instnorm_nhwc")
    return y


def maxpool2x2(x) -> torch.Tensor:
    _chk_any(x)
    n, c, h, w = x.shape
    y = empty_nhwc(n, c, h // 2, w // 2, x.device, x.dtype)
    with torch.cuda.device(x.device):
        capi.check(capi.lib().ie_maxpool2x2_nhwc(capi.ptr(x), capi.ptr(y), _dt(x), n, h, w, c, capi.current_stream()), "ie_maxpool2x2_nhwc")
    return y


_RESIZE = {"nearest": 0, "bilinear": 1, "bilinear_align": 2}


def resize(x, size, mode="bilinear_align") -> torch.Tensor:
    _chk_any(x)
    n, c, h, w = x.shape
    y = empty_nhwc(n, c, size[0], size[1], x.device, x.dtype)
    with torch.cuda.device(x.device):
        capi.check(capi.lib().ie_resize_nhwc(capi.ptr(x), capi.ptr(y), _dt(x), n, h, w, c, size[0], size[1], _RESIZE[mode],
                                             capi.current_stream()), "ie_resize_nhwc")
    return y


def resize_pad(x, size, pads, mode="bilinear_align") -> torch.Tensor:
    """``F.pad(resize(x, size), [left, right, top, bottom])`` in one pass on a 16-bit tensor (ie_resize_pad_nhwc); ``pads`` = (top, bottom, left, right)."""
    _chk16(x)
    n, c, h, w = x.shape
    top, bottom, left, right = (int(v) for v in pads)
    y = empty_nhwc(n, c, size[0] + top + bottom, size[1] + left + right, x.device, x.dtype)
    with torch.cuda.device(x.device):
        capi.check(capi.lib().ie_resize_pad_nhwc(capi.ptr(x), capi.ptr(y), _dt(x), n, h, w, c, size[0], size[1], _RESIZE[mode], top, bottom, left, right,
                                                 capi.current_stream()), "ie_resize_pad_nhwc")
    return y


def upsample2x_bilinear(x) -> torch.Tensor:
    return resize(x, (x.shape[2] * 2, x.shape[3] * 2), "bilinear_align")


def pixel_shuffle(x, r=2) -> torch.Tensor:
    _chk_any(x)
    n, c, h, w = x.shape
    co = c // (r * r)
    y = empty_nhwc(n, co, h * r, w * r, x.device, x.dtype)
    with torch.cuda.device(x.device):
        capi.check(capi.lib().ie_pixel_shuffle_nhwc(capi.ptr(x), capi.ptr(y), _dt(x), n, h, w, co, r, capi.current_stream()),
                   "ie_pixel_shuffle_nhwc")
    return y


def cast(x, dtype, channels=None) -> torch.Tensor:
    """Convert dtype and pad (zeros) / crop the channel count in one pass (ie_cast_nhwc)."""
    _chk_any(x)
    n, c, h, w = x.shape
    cd = c if channels is None else int(channels)
    y = empty_nhwc(n, cd, h, w, x.device, dtype)
    with torch.cuda.device(x.device):
        capi.check(capi.lib().ie_cast_nhwc(capi.ptr(x), _dt(x), c, capi.ptr(y), _dt(y), cd, n * h * w, capi.current_stream()), "ie_cast_nhwc")
    return y


def concat_cast(srcs, dtype, channels) -> torch.Tensor:
    """``torch.cat(srcs, 1)`` of fp32 NHWC tensors, converted to ``dtype`` and zero-padded to ``channels`` channels in one pass (ie_concat_cast_nhwc)."""
    for t in srcs:
        _chk(t)
    n, _, h, w = srcs[0].shape
    if any(tuple(t.shape[0:1] + t.shape[2:]) != (n, h, w) for t in srcs):
        raise ValueError("concat_cast: sources must share N, H, W")
    y = empty_nhwc(n, int(channels), h, w, srcs[0].device, dtype)
    ptrs = (C.c_void_p * len(srcs))(*[capi.ptr(t) for t in srcs])
    chans = (C.c_int * len(srcs))(*[int(t.shape[1]) for t in srcs])
    with torch.cuda.device(y.device):
        capi.check(capi.lib().ie_concat_cast_nhwc(ptrs, chans, len(srcs), capi.ptr(y), _dt(y), int(channels), n * h * w, capi.current_stream()),
                   "ie_concat_cast_nhwc")
    return y


class ConvT2x2Weights:
    def __init__(self, w_iohw: torch.Tensor, bias, device):
        w = w_iohw.detach().to(torch.float32)                 # (Cin, Cout, 2, 2)
        self.cin, self.cout = w.shape[0], w.shape[1]
        self.w = w.permute(2, 3, 0, 1).contiguous().to(device)   # [i][j][Cin][Cout]; host-side repack
        self.b = None if bias is None else bias.detach().to(device, torch.float32).contiguous()


def conv_transpose2x2(x, cw: ConvT2x2Weights) -> torch.Tensor:
    _chk(x)
    n, c, h, w = x.shape
    y = empty_nhwc(n, cw.cout, 2 * h, 2 * w, x.device)
    with torch.cuda.device(x.device):
        capi.check(capi.lib().ie_conv_transpose2x2_nhwc_f32(capi.ptr(x), capi.ptr(cw.w), capi.ptr(cw.b), capi.ptr(y), n, h, w, c,
                                                            cw.cout, capi.current_stream()), "ie_conv_transpose2x2_nhwc_f32")
    return y


def pad(x, top, bottom, left, right) -> torch.Tensor:
    _chk_any(x)
    if top == bottom == left == right == 0:
        return x
    n, c, h, w = x.shape
    y = empty_nhwc(n, c, h + top + bottom, w + left + right, x.device, x.dtype)
    with torch.cuda.device(x.device):
        capi.check(capi.lib().ie_pad_nhwc(capi.ptr(x), capi.ptr(y), _dt(x), n, h, w, c, top, bottom, left, right, capi.current_stream()),
                   "ie_pad_nhwc")
    return y


def _elt(op, a, b, lo=0.0, hi=0.0):
    if a.dtype != torch.float32:
        _chk16(a); _chk16(b)
        n, c, h, w = a.shape
        y = torch.empty_like(a, memory_format=CL)
        with torch.cuda.device(a.device):
            capi.check(capi.lib().ie_eltwise_nhwc(op, capi.ptr(a), capi.ptr(b), capi.ptr(y), _dt(a), n * h * w, c, int(b.shape[1]),
                                                  capi.current_stream()), "ie_eltwise_nhwc")
        return y
    _chk(a)
    n, c, h, w = a.shape
    cb = c
    if b is not None:
        _chk(b)
        cb = int(b.shape[1])
    y = torch.empty_like(a, memory_format=CL)
    with torch.cuda.device(a.device):
        capi.check(capi.lib().ie_eltwise_nhwc_f32(op, capi.ptr(a), capi.ptr(b), capi.ptr(y), n * h * w, c, cb, lo, hi,
                                                  capi.current_stream()), "ie_eltwise_nhwc_f32")
    return y


def add(a, b):
    return _elt(0, a, b)


def mul(a, b):
    """a * b, where b has the same channels as a or a single channel (broadcast)."""
    return _elt(1, a, b)


def clamp(a, lo, hi):
    return _elt(2, a, None, lo, hi)


def sub(a, b):
    return _elt(4, a, b)


def div_clamp(a, b, lo=float("-inf"), hi=float("inf")):
    """clamp(a / b, lo, hi); b has the channels of a or one channel."""
    return _elt(5, a, b, lo, hi)


def affine(a, scale=1.0, shift=0.0):
    """a * scale + shift (fp32)."""
    return _elt(6, a, None, scale, shift)


def pow_scalar(a, exponent):
    return _elt(7, a, None, exponent, 0.0)


def channel_stats(x) -> torch.Tensor:
    """(N, C, 2) fp64 {sum, sum of squares} over H*W of an fp32 / 16-bit NHWC tensor."""
    _chk_any(x)
    n, c, h, w = x.shape
    st = torch.empty((n, c, 2), device=x.device, dtype=torch.float64)
    with torch.cuda.device(x.device):
        capi.check(capi.lib().ie_channel_stats_nhwc(capi.ptr(x), _dt(x), capi.ptr(st), n, h, w, c, capi.current_stream()), "ie_channel_stats_nhwc")
    return st


def channel_affine(x, scale, shift=None, act="none", slope=0.0, residual=None) -> torch.Tensor:
    """act(x * scale[n, c] + shift[n, c]) (+ residual, same pass, one rounding) with (N, C) fp32 coefficient tensors; x fp32 or 16-bit
    (result in x's type)."""
    _chk_any(x)
    n, c, h, w = x.shape
    if residual is not None:
        _chk_any(residual)
        if residual.dtype != x.dtype or tuple(residual.shape) != tuple(x.shape):
            raise capi.IEError("channel_affine: residual must match x")
    y = torch.empty_like(x, memory_format=CL)
    with torch.cuda.device(x.device):
        capi.check(capi.lib().ie_channel_affine_add_nhwc(capi.ptr(x), capi.ptr(scale), capi.ptr(shift), capi.ptr(residual), capi.ptr(y), _dt(x), n, h, w, c,
                                                         capi.ACT[act], slope, capi.current_stream()), "ie_channel_affine_add_nhwc")
    return y


def channel_affine_act(x, act, slope=0.0) -> torch.Tensor:
    """Plain activation as its own pass (x * 1 + 0 per channel): LeakyReLU after a conv whose norm is nn.Identity (lllinet.py:113-115)."""
    n, c, h, w = x.shape
    one = torch.ones((n, c), device=x.device, dtype=torch.float32)
    return channel_affine(x, one, None, act, slope)


def learnable_instance_norm(x, gamma, beta, r, eps=1e-5, act="none", slope=0.0) -> torch.Tensor:
    """LearnableInstanceNorm2d (normalization.py:177-224) + activation: IN(x) * r + x * (1 - r)."""
    n, c, h, w = x.shape
    st = channel_stats(x)
    scale = torch.empty((n, c), device=x.device, dtype=torch.float32)
    shift = torch.empty_like(scale)
    with torch.cuda.device(x.device):
        capi.check(capi.lib().ie_lin_coeffs_f32(capi.ptr(st), capi.ptr(gamma), capi.ptr(beta), capi.ptr(r), capi.ptr(scale), capi.ptr(shift), n, c, h * w, eps,
                                                capi.current_stream()), "ie_lin_coeffs_f32")
    return channel_affine(x, scale, shift, act, slope)


def simam(x, e_lambda=1e-4) -> torch.Tensor:
    _chk_any(x)
    n, c, h, w = x.shape
    st = channel_stats(x)
    y = torch.empty_like(x, memory_format=CL)
    with torch.cuda.device(x.device):
        capi.check(capi.lib().ie_simam_nhwc(capi.ptr(x), capi.ptr(st), capi.ptr(y), _dt(x), n, h, w, c, e_lambda, capi.current_stream()), "ie_simam_nhwc")
    return y


def global_avg_pool(x) -> torch.Tensor:
    """nn.AdaptiveAvgPool2d(1): (N, C, 1, 1) fp32 NHWC."""
    n, c, h, w = x.shape
    st = channel_stats(x)
    m = empty_nhwc(n, c, 1, 1, x.device)
    with torch.cuda.device(x.device):
        capi.check(capi.lib().ie_stats_mean_f32(capi.ptr(st), capi.ptr(m), n, c, h * w, capi.current_stream()), "ie_stats_mean_f32")
    return m


def div_by_max(a) -> torch.Tensor:
    """a / max(a) over the whole tensor (zero_dce_v.py:205); the maximum stays on the device."""
    _chk(a)
    m = torch.empty(1, device=a.device, dtype=torch.float32)
    ws = torch.empty(1024, device=a.device, dtype=torch.float32)
    y = torch.empty_like(a, memory_format=CL)
    n, c, h, w = a.shape
    with torch.cuda.device(a.device):
        capi.check(capi.lib().ie_reduce_max_f32(capi.ptr(a), a.numel(), capi.ptr(m), capi.ptr(ws), capi.current_stream()), "ie_reduce_max_f32")
        capi.check(capi.lib().ie_eltwise_nhwc_f32(8, capi.ptr(a), capi.ptr(m), capi.ptr(y), n * h * w, c, 1, 0.0, 0.0, capi.current_stream()),
                   "ie_eltwise_nhwc_f32")
    return y


def threshold_of_max(a, thr) -> torch.Tensor:
    """(a / max(a) > thr).float() (BoundaryAwarePrior, nn/modules/prior/image.py:205-208)."""
    _chk(a)
    m = torch.empty(1, device=a.device, dtype=torch.float32)
    ws = torch.empty(1024, device=a.device, dtype=torch.float32)
    y = torch.empty_like(a, memory_format=CL)
    n, c, h, w = a.shape
    with torch.cuda.device(a.device):
        capi.check(capi.lib().ie_reduce_max_f32(capi.ptr(a), a.numel(), capi.ptr(m), capi.ptr(ws), capi.current_stream()), "ie_reduce_max_f32")
        capi.check(capi.lib().ie_eltwise_nhwc_f32(9, capi.ptr(a), capi.ptr(m), capi.ptr(y), n * h * w, c, 1, thr, 0.0, capi.current_stream()),
                   "ie_eltwise_nhwc_f32")
    return y


def rgb_to_hsv(x, eps=1e-8, want_hsv=True, want_v=True):
    """core.rgb_to_hsv / core.rgb_to_v on an fp32 NHWC RGB tensor -> (hsv (N,3,H,W) or None, v (N,1,H,W) or None)."""
    _chk(x)
    n, c, h, w = x.shape
    if c != 3:
        raise ValueError(f"rgb_to_hsv needs 3 channels, got {c}")
    hsv = empty_nhwc(n, 3, h, w, x.device) if want_hsv else None
    v = empty_nhwc(n, 1, h, w, x.device) if want_v else None
    with torch.cuda.device(x.device):
        capi.check(capi.lib().ie_rgb_to_hsv_nhwc_f32(capi.ptr(x), capi.ptr(hsv), capi.ptr(v), n * h * w, eps, capi.current_stream()), "ie_rgb_to_hsv_nhwc_f32")
    return hsv, v


def rgb_to_hvi(x, density_k=0.2, eps=1e-8):
    """core.RGBToHVI.rgb_to_hvi on an fp32 NHWC RGB tensor (ie_rgb_to_hvi_nhwc_f32)."""
    _chk(x)
    n, c, h, w = x.shape
    if c != 3:
        raise ValueError(f"rgb_to_hvi needs 3 channels, got {c}")
    y = empty_nhwc(n, 3, h, w, x.device)
    with torch.cuda.device(x.device):
        capi.check(capi.lib().ie_rgb_to_hvi_nhwc_f32(capi.ptr(x), capi.ptr(y), n * h * w, float(density_k), float(eps), capi.current_stream()), "ie_rgb_to_hvi_nhwc_f32")
    return y


def hvi_to_rgb(x, density_k=0.2, eps=1e-8, clamp01=False):
    """core.RGBToHVI.hvi_to_rgb (+ optional clamp to [0, 1]) on an fp32 NHWC tensor (ie_hvi_to_rgb_nhwc_f32)."""
    _chk(x)
    n, c, h, w = x.shape
    if c != 3:
        raise ValueError(f"hvi_to_rgb needs 3 channels, got {c}")
    y = empty_nhwc(n, 3, h, w, x.device)
    with torch.cuda.device(x.device):
        capi.check(capi.lib().ie_hvi_to_rgb_nhwc_f32(capi.ptr(x), capi.ptr(y), n * h * w, float(density_k), float(eps), int(bool(clamp01)), capi.current_stream()),
                   "ie_hvi_to_rgb_nhwc_f32")
    return y


def hsv_to_rgb(hsv, v_new=None):
    _chk(hsv)
    n, c, h, w = hsv.shape
    y = torch.empty_like(hsv, memory_format=CL)
    with torch.cuda.device(hsv.device):
        capi.check(capi.lib().ie_hsv_to_rgb_nhwc_f32(capi.ptr(hsv), capi.ptr(v_new), capi.ptr(y), n * h * w, capi.current_stream()), "ie_hsv_to_rgb_nhwc_f32")
    return y


def le_curve(x, adjust, iters, bam=None):
    """LE curve on an fp32 NHWC image of C channels with C*iters (or C: shared) curve maps; ``bam``: GCENet's attention form."""
    _chk(x); _chk(adjust)
    n, c, h, w = x.shape
    y = torch.empty_like(x, memory_format=CL)
    with torch.cuda.device(x.device):
        capi.check(capi.lib().ie_le_curve_nhwc_f32(capi.ptr(x), capi.ptr(adjust), capi.ptr(bam), capi.ptr(y), n * h * w, c, int(adjust.shape[1]),
                                                   iters, capi.current_stream()), "ie_le_curve_nhwc_f32")
    return y


def box_filter(x, radius):
    _chk(x)
    n, c, h, w = x.shape
    y = torch.empty_like(x, memory_format=CL)
    with torch.cuda.device(x.device):
        capi.check(capi.lib().ie_box_filter_nhwc_f32(capi.ptr(x), capi.ptr(y), n, h, w, c, radius, capi.current_stream()), "ie_box_filter_nhwc_f32")
    return y


def guided_filter_module(x, y, radius, eps=1e-8):
    """``GuidedFilter(radius, eps)(image=x, guide=y)`` as the reference CLASS behaves (guided_filter.py:111-146): it hands ``radius`` to BoxFilter as the kernel SIZE
    (box_filter.py:163-172), so its windows are ``radius`` pixels wide = box radius int((radius - 1) / 2); radius 1 / 2 give the guide back (window of one pixel:
    cov = 0, A = 0, b = y).  Found by the fixture tests/golden/guided_filter_seed161.npz; GCENet's gf_radius = 3 means 3x3 windows."""
    rb = int((radius - 1) / 2)
    if rb < 1:
        _chk(x); _chk(y)
        return y.clone(memory_format=CL)
    return guided_filter(x, y, rb, eps)


def guided_filter(x, y, radius, eps=1e-8):
    """The guided filter with (2 radius + 1)^2 windows (He et al.; the arithmetic of guided_filter.py:117-146 with a BoxFilter of that radius).  The reference's
    GuidedFilter class does NOT use its ``radius`` this way: see ``guided_filter_module``."""
    _chk(x); _chk(y)
    n, cy, h, w = y.shape
    out = torch.empty_like(y, memory_format=CL)
    ws = torch.empty(2 * y.numel(), device=y.device, dtype=torch.float32)
    with torch.cuda.device(y.device):
        capi.check(capi.lib().ie_guided_filter_nhwc_f32(capi.ptr(x), capi.ptr(y), capi.ptr(out), capi.ptr(ws), n, h, w, int(x.shape[1]), cy, radius, eps,
                                                        capi.current_stream()), "ie_guided_filter_nhwc_f32")
    return out


def fast_guided_filter(x_lr, y_lr, x_hr, radius, eps=1e-8, clip=None):
    """FastGuidedFilter(radius, eps)(x_lr, y_lr, x_hr) (guided_filter.py:150-204), optionally clipped."""
    _chk(x_lr); _chk(y_lr); _chk(x_hr)
    n, cy, h, w = y_lr.shape
    H, W = int(x_hr.shape[2]), int(x_hr.shape[3])
    out = empty_nhwc(n, cy, H, W, x_hr.device)
    ws = torch.empty(2 * y_lr.numel(), device=y_lr.device, dtype=torch.float32)
    lo, hi = clip if clip is not None else (0.0, 0.0)
    with torch.cuda.device(x_hr.device):
        capi.check(capi.lib().ie_fast_guided_filter_nhwc_f32(capi.ptr(x_lr), capi.ptr(y_lr), capi.ptr(x_hr), capi.ptr(out), capi.ptr(ws), n, h, w, H, W,
                                                             int(x_lr.shape[1]), cy, radius, eps, 1 if clip is not None else 0, lo, hi,
                                                             capi.current_stream()), "ie_fast_guided_filter_nhwc_f32")
    return out


def resize_bicubic_align(x, size):
    _chk(x)
    n, c, h, w = x.shape
    y = empty_nhwc(n, c, size[0], size[1], x.device)
    with torch.cuda.device(x.device):
        capi.check(capi.lib().ie_resize_bicubic_nhwc_f32(capi.ptr(x), capi.ptr(y), n, h, w, size[0], size[1], c, capi.current_stream()), "ie_resize_bicubic_nhwc_f32")
    return y


def median_blur(x, ksize):
    _chk(x)
    n, c, h, w = x.shape
    y = torch.empty_like(x, memory_format=CL)
    with torch.cuda.device(x.device):
        capi.check(capi.lib().ie_median_blur_nhwc_f32(capi.ptr(x), capi.ptr(y), n, h, w, c, ksize, capi.current_stream()), "ie_median_blur_nhwc_f32")
    return y


def sobel(x, eps=1e-6):
    _chk(x)
    n, c, h, w = x.shape
    y = torch.empty_like(x, memory_format=CL)
    with torch.cuda.device(x.device):
        capi.check(capi.lib().ie_sobel_nhwc_f32(capi.ptr(x), capi.ptr(y), n, h, w, c, eps, capi.current_stream()), "ie_sobel_nhwc_f32")
    return y


def mul_prefix(a, b) -> torch.Tensor:
    """y = a[:, :C] * b for a with C+1 channels and a single-channel b (RetinexNet: S = R * I_delta)."""
    _chk(a); _chk(b)
    n, c1, h, w = a.shape
    y = empty_nhwc(n, c1 - 1, h, w, a.device)
    with torch.cuda.device(a.device):
        capi.check(capi.lib().ie_eltwise_nhwc_f32(3, capi.ptr(a), capi.ptr(b), capi.ptr(y), n * h * w, c1 - 1, 1, 0.0, 0.0,
                                                  capi.current_stream()), "ie_eltwise_nhwc_f32")
    return y


def max_concat(x) -> torch.Tensor:
    """[max over channels, x] (RetinexNet DecomNet input)."""
    _chk(x)
    n, c, h, w = x.shape
    y = empty_nhwc(n, c + 1, h, w, x.device)
    with torch.cuda.device(x.device):
        capi.check(capi.lib().ie_max_concat_nhwc_f32(capi.ptr(x), capi.ptr(y), n * h * w, c, capi.current_stream()),
                   "ie_max_concat_nhwc_f32")
    return y


# ---- training path: norm layers with saved statistics, backward ops (include/ie_hip.h "training path") -------------
class ConvWeightsView(ConvWeights):
    """ConvWeights over caller-owned device storage (a slice of the trainer's flat parameter buffer, HWIO)."""

    def __init__(self, w_flat: torch.Tensor, bias, cout: int, cin_g: int, kh: int, kw: int):
        self.w, self.b, self.cout, self.cin_g, self.kh, self.kw = w_flat, bias, cout, cin_g, kh, kw


def pack_hwio_into(w_oihw: torch.Tensor, dst_flat: torch.Tensor):
    """Checkpoint OIHW (host) -> HWIO into ``dst_flat`` (device view)."""
    w = w_oihw.detach().to("cpu", torch.float32).contiguous()
    co, ci, kh, kw = w.shape
    with torch.cuda.device(dst_flat.device):
        capi.check(capi.lib().ie_pack_weight_hwio_f32(capi.ptr(w), co, ci, kh, kw, capi.ptr(dst_flat), capi.current_stream()),
                   "ie_pack_weight_hwio_f32")
        torch.cuda.current_stream().synchronize()      # `w` is a temporary host buffer


def norm_forward(x, gamma, beta, per_sample: bool, running_mean=None, running_var=None, momentum=0.1, eps=1e-5, act="none",
                 slope=0.0, c_begin=0, c_count=None):
    """InstanceNorm2d (per_sample) / training-mode BatchNorm2d; returns (y, save_mean, save_invstd)."""
    _chk_any(x)
    n, c, h, w = x.shape
    c_count = c - c_begin if c_count is None else c_count
    g = n if per_sample else 1
    y = torch.empty_like(x, memory_format=CL)
    ws = torch.empty(2 * g * c_count, device=x.device, dtype=torch.float64)
    mean = torch.empty(g * c_count, device=x.device, dtype=torch.float32)
    inv = torch.empty_like(mean)
    with torch.cuda.device(x.device):
        capi.check(capi.lib().ie_norm_forward_nhwc(capi.ptr(x), capi.ptr(gamma), capi.ptr(beta), capi.ptr(y), _dt(x), capi.ptr(ws),
                                                   capi.ptr(mean), capi.ptr(inv), capi.ptr(running_mean), capi.ptr(running_var),
                                                   momentum, int(per_sample), n, h, w, c, c_begin, c_count, eps, capi.ACT[act],
                                                   slope, capi.current_stream()), "ie_norm_forward_nhwc")
    return y, mean, inv


def norm_backward(x, dy, y, gamma, mean, inv, per_sample: bool, dgamma=None, dbeta=None, accumulate=False, act="none", slope=0.0,
                  c_begin=0, c_count=None, beta=None, recompute=False):
    """``recompute`` (act none / relu / lrelu): the activation's derivative comes from the recomputed pre-activation gamma * xhat + beta instead of the
    stored output ``y`` (ie_norm_backward_pre_nhwc: neither pass reads ``y``; pass ``beta`` whenever ``gamma`` is given)."""
    _chk_any(x); _chk_any(dy)
    if dy.dtype != x.dtype:
        raise capi.IEError("norm_backward: x and dy must have the same dtype")
    n, c, h, w = x.shape
    c_count = c - c_begin if c_count is None else c_count
    g = n if per_sample else 1
    dx = torch.empty_like(x, memory_format=CL)
    ws = torch.empty(2 * g * c_count, device=x.device, dtype=torch.float64)
    if recompute:
        with torch.cuda.device(x.device):
            capi.check(capi.lib().ie_norm_backward_pre_nhwc(capi.ptr(x), capi.ptr(dy), _dt(x), capi.ptr(gamma), capi.ptr(beta), capi.ptr(mean),
                                                            capi.ptr(inv), capi.ptr(ws), capi.ptr(dx), capi.ptr(dgamma), capi.ptr(dbeta),
                                                            int(accumulate), int(per_sample), n, h, w, c, c_begin, c_count,
                                                            capi.ACT[act], slope, capi.current_stream()), "ie_norm_backward_pre_nhwc")
        return dx
    with torch.cuda.device(x.device):
        capi.check(capi.lib().ie_norm_backward_nhwc(capi.ptr(x), capi.ptr(dy), capi.ptr(y), _dt(x), capi.ptr(gamma), capi.ptr(mean),
                                                    capi.ptr(inv), capi.ptr(ws), capi.ptr(dx), capi.ptr(dgamma), capi.ptr(dbeta),
                                                    int(accumulate), int(per_sample), n, h, w, c, c_begin, c_count,
                                                    capi.ACT[act], slope, capi.current_stream()), "ie_norm_backward_nhwc")
    return dx


def _convn_desc(srcs, cw, stride, pad, pad_mode):
    n, _, h, w = srcs[0].shape
    return capi.ConvNDesc(n, h, w, len(srcs), cw.cout, cw.kh, cw.kw, stride, pad, capi.PAD[pad_mode], 0, 0.0, 0, 0.0, 0.0, 0, 0)


def conv2d_wgrad(srcs, dy, cw: ConvWeights, dw: torch.Tensor, dbias=None, stride=1, pad=1, pad_mode="zeros", accumulate=False):
    """dW (HWIO, flat) and optional dbias of conv2d(srcs, cw) given the gradient ``dy`` at its output."""
    if isinstance(srcs, torch.Tensor):
        srcs = [srcs]
    for s in srcs:
        _chk_any(s)
        if s.dtype != dy.dtype:
            raise capi.IEError("conv2d_wgrad: sources and dy must have the same dtype")
    _chk_any(dy)
    d = _convn_desc(srcs, cw, stride, pad, pad_mode)
    cs = [int(s.shape[1]) for s in srcs]
    if dw.numel() != cw.kh * cw.kw * sum(cs) * cw.cout:
        raise capi.IEError("conv2d_wgrad: dw has the wrong size")
    ptrs = (C.c_void_p * len(srcs))(*[s.data_ptr() for s in srcs])
    chans = (C.c_int * len(srcs))(*cs)
    ws = torch.empty(cw.cout, device=dy.device, dtype=torch.float64) if dbias is not None else None
    with torch.cuda.device(dy.device):
        capi.check(capi.lib().ie_convn_wgrad_nhwc(C.byref(d), ptrs, chans, capi.ptr(dy), _dt(dy), capi.ptr(dw), capi.ptr(dbias),
                                                  int(accumulate), capi.ptr(ws), capi.current_stream()), "ie_convn_wgrad_nhwc")


def conv2d_dgrad(dy, cw: ConvWeights, src_shapes, dsts, accumulate, stride=1, pad=1, pad_mode="zeros"):
    """Data gradients of conv2d over a virtual concat: ``dsts[i]`` (or None) receives the gradient of source i."""
    _chk(dy)
    n, _, h, w = src_shapes[0]
    cs = [int(s[1]) for s in src_shapes]
    d = capi.ConvNDesc(n, h, w, len(cs), cw.cout, cw.kh, cw.kw, stride, pad, capi.PAD[pad_mode], 0, 0.0, 0, 0.0, 0.0, 0, 0)
    ptrs = (C.c_void_p * len(cs))(*[None if t is None else t.data_ptr() for t in dsts])
    chans = (C.c_int * len(cs))(*cs)
    acc = (C.c_int * len(cs))(*[int(a) for a in accumulate])
    with torch.cuda.device(dy.device):
        capi.check(capi.lib().ie_convn_dgrad_nhwc_f32(C.byref(d), capi.ptr(dy), capi.ptr(cw.w), ptrs, chans, acc,
                                                      capi.current_stream()), "ie_convn_dgrad_nhwc_f32")


def maxpool2x2_backward(x, dy, dx, accumulate):
    _chk_any(x); _chk_any(dy); _chk_any(dx)
    n, c, h, w = x.shape
    with torch.cuda.device(x.device):
        capi.check(capi.lib().ie_maxpool2x2_backward_nhwc(capi.ptr(x), capi.ptr(dy), capi.ptr(dx), _dt(x), n, h, w, c, int(accumulate),
                                                          capi.current_stream()), "ie_maxpool2x2_backward_nhwc")


def resize_backward(dy, dx, mode, accumulate):
    _chk_any(dy); _chk_any(dx)
    n, c, hi, wi = dx.shape
    with torch.cuda.device(dy.device):
        capi.check(capi.lib().ie_resize_backward_nhwc(capi.ptr(dy), capi.ptr(dx), _dt(dy), n, hi, wi, c, dy.shape[2], dy.shape[3],
                                                      _RESIZE[mode], int(accumulate), capi.current_stream()),
                   "ie_resize_backward_nhwc")


def crop_into(src, dst, top, left, accumulate):
    _chk_any(src); _chk_any(dst)
    n, c, h, w = dst.shape
    with torch.cuda.device(src.device):
        capi.check(capi.lib().ie_crop_nhwc(capi.ptr(src), capi.ptr(dst), _dt(src), n, src.shape[2], src.shape[3], c, top, left, h, w,
                                           int(accumulate), capi.current_stream()), "ie_crop_nhwc")


def pad_replicate(x, pad) -> torch.Tensor:
    """F.pad(x, (pad,)*4, mode="replicate") on an NHWC tensor (fp32, or 16-bit with a multiple of 8 channels)."""
    _chk_any(x)
    if pad == 0:
        return x
    n, c, h, w = x.shape
    y = empty_nhwc(n, c, h + 2 * pad, w + 2 * pad, x.device, x.dtype)
    with torch.cuda.device(x.device):
        capi.check(capi.lib().ie_pad_replicate_nhwc(capi.ptr(x), capi.ptr(y), _dt(x), n, h, w, c, pad, capi.current_stream()), "ie_pad_replicate_nhwc")
    return y


def pad_replicate_backward_into(dy, dx, pad, accumulate):
    _chk_any(dy); _chk_any(dx)
    if dy.dtype != dx.dtype:
        raise capi.IEError("pad_replicate_backward_into: dy and dx must share a dtype")
    n, c, h, w = dx.shape
    with torch.cuda.device(dy.device):
        capi.check(capi.lib().ie_pad_replicate_backward_nhwc(capi.ptr(dy), capi.ptr(dx), _dt(dy), n, h, w, c, pad, int(accumulate), capi.current_stream()),
                   "ie_pad_replicate_backward_nhwc")


def zero_insert2(x, ho, wo) -> torch.Tensor:
    """dst[n, :, 2h, 2w] = x[n, :, h, w], zeros elsewhere, dst (N, C, ho, wo): the data gradient of a stride-2 convolution is a stride-1 convolution of this."""
    _chk16(x)
    n, c, h, w = (int(v) for v in x.shape)
    y = torch.empty((n, c, int(ho), int(wo)), device=x.device, dtype=x.dtype, memory_format=CL)
    with torch.cuda.device(x.device):
        capi.check(capi.lib().ie_zero_insert2_nhwc(capi.ptr(x), capi.ptr(y), IE_DT[x.dtype], n, h, w, c, int(ho), int(wo), capi.current_stream()), "ie_zero_insert2_nhwc")
    return y


def crop(x, top, left, h, w) -> torch.Tensor:
    """x[:, :, top:top+h, left:left+w] as its own NHWC tensor."""
    y = empty_nhwc(x.shape[0], x.shape[1], h, w, x.device, x.dtype)
    crop_into(x, y, top, left, False)
    return y


def cat_spatial(tiles, dim) -> torch.Tensor:
    """torch.cat(tiles, dim) for dim 2 (H) or 3 (W) of NHWC tensors with equal channel counts: every tile is pasted into its place."""
    n, c = tiles[0].shape[0], tiles[0].shape[1]
    hs, ws = [t.shape[2] for t in tiles], [t.shape[3] for t in tiles]
    H, W = (sum(hs), ws[0]) if dim == 2 else (hs[0], sum(ws))
    y = empty_nhwc(n, c, H, W, tiles[0].device, tiles[0].dtype)
    off = 0
    with torch.cuda.device(y.device):
        for t in tiles:
            _chk_any(t)
            top, left = (off, 0) if dim == 2 else (0, off)
            capi.check(capi.lib().ie_paste_nhwc(capi.ptr(t), capi.ptr(y), _dt(t), n, t.shape[2], t.shape[3], c, H, W, top, left, capi.current_stream()),
                       "ie_paste_nhwc")
            off += t.shape[2] if dim == 2 else t.shape[3]
    return y


def pixel_unshuffle_into(dy, dx, r, accumulate):
    _chk_any(dy); _chk_any(dx)
    if dy.dtype != dx.dtype:
        raise capi.IEError("pixel_unshuffle_into: dy and dx must share a dtype")
    n, c, h, w = dx.shape
    with torch.cuda.device(dy.device):
        capi.check(capi.lib().ie_pixel_unshuffle_nhwc(capi.ptr(dy), capi.ptr(dx), _dt(dy), n, h, w, c // (r * r), r, int(accumulate),
                                                      capi.current_stream()), "ie_pixel_unshuffle_nhwc")


def act_backward_into(dy, ref, dx, act, slope=0.0, lo=0.0, hi=0.0, accumulate=False):
    """dx (+)= dy * act'(ref); act in ACT names (ref = activation output) or "clamp" (ref = clamp input)."""
    code = 5 if act == "clamp" else capi.ACT[act]
    if not (dy.dtype == ref.dtype == dx.dtype):
        raise capi.IEError("act_backward_into: dy, ref and dx must share a dtype")
    with torch.cuda.device(dy.device):
        capi.check(capi.lib().ie_act_backward(capi.ptr(dy), capi.ptr(ref), capi.ptr(dx), _dt(dy), dy.numel(), code, slope, lo, hi,
                                              int(accumulate), capi.current_stream()), "ie_act_backward")


# ---- 16-bit matrix-core path (ie_mconv_*) ------------------------------------------------------------------------------
TORCH_DT = {"f16": torch.float16, "bf16": torch.bfloat16, "f32": torch.float32}
IE_DT = {torch.float16: capi.IE_F16, torch.bfloat16: capi.IE_BF16, torch.float32: capi.IE_F32}


def _chk16(t: torch.Tensor) -> torch.Tensor:
    if not t.is_cuda or t.dtype not in (torch.float16, torch.bfloat16) or t.dim() != 4:
        raise capi.IEError(f"expected a cuda fp16/bf16 (N,C,H,W) tensor, got {tuple(t.shape)} {t.dtype} {t.device}")
    if not t.is_contiguous(memory_format=CL):
        raise capi.IEError("expected channels_last (NHWC) memory")
    return t


def to_nhwc16(x: torch.Tensor, dtype, cpad: int = 8) -> torch.Tensor:
    """NCHW fp32 image -> channels_last 16-bit with the channel count padded to a multiple of ``cpad`` (zeros)."""
    x = x.to(torch.float32).contiguous()
    n, c, h, w = x.shape
    cp = (c + cpad - 1) // cpad * cpad
    y = torch.empty((n, cp, h, w), device=x.device, dtype=dtype, memory_format=CL)
    with torch.cuda.device(x.device):
        capi.check(capi.lib().ie_nchw_f32_to_nhwc(capi.ptr(x), capi.ptr(y), IE_DT[dtype], n, c, h, w, cp, capi.current_stream()),
                   "ie_nchw_f32_to_nhwc")
    return y


class MConvWeights:
    """Weights of one Conv2d repacked for ie_mconv_nhwc: MFMA A-operand fragments per (cout tile, K chunk, tap), K chunked
    per source of the virtual concat.  ``split``: channel counts of the sources as they will be passed (each a multiple
    of 8; weights of padded input channels are zero).  ``cin_map``: for every source, the indices (into the checkpoint's
    Cin) its channels correspond to, or -1 for padding channels."""

    def __init__(self, w_oihw: torch.Tensor, bias, split, dtype, device, cin_map=None, shuffle: int = 0):
        w = w_oihw.detach().to("cpu", torch.float32)
        self.shuffle = int(shuffle)
        if self.shuffle:                  # fused PixelShuffle(2): output rows in [i][j][c] order (row (2i + j) C + c <- checkpoint row 4c + 2i + j)
            if self.shuffle != 2 or w.shape[0] % 4:
                raise capi.IEError("MConvWeights: fused pixel shuffle needs r = 2 and Cout divisible by 4")
            c4 = w.shape[0] // 4
            perm = torch.arange(w.shape[0]).reshape(c4, 4).t().reshape(-1)
            w = w[perm]
            bias = None if bias is None else bias.detach().to("cpu", torch.float32)[perm]
        self.cout, cin, self.kh, self.kw = w.shape
        self.split = [int(s) for s in split]
        if cin_map is not None:
            idx = torch.as_tensor(cin_map, dtype=torch.long)
            wz = torch.zeros(self.cout, idx.numel(), self.kh, self.kw)
            wz[:, idx >= 0] = w[:, idx[idx >= 0]]
            w = wz
        if w.shape[1] != sum(self.split):
            raise capi.IEError(f"MConvWeights: weights carry {w.shape[1]} input channels, sources {sum(self.split)}")
        w = w.contiguous()
        self.dtype = dtype
        chans = (C.c_int * len(self.split))(*self.split)
        L = capi.lib()
        nbytes = L.ie_mconv_packed_bytes(chans, len(self.split), self.cout, self.kh, self.kw)
        if nbytes == 0:
            capi.check(-1, "ie_mconv_packed_bytes")
        self.w = torch.empty(nbytes, device=device, dtype=torch.uint8)
        with torch.cuda.device(device):
            capi.check(L.ie_mconv_pack_weights(capi.ptr(w), chans, len(self.split), self.cout, self.kh, self.kw, IE_DT[dtype],
                                               capi.ptr(self.w), capi.current_stream()), "ie_mconv_pack_weights")
        self.b = None if bias is None else bias.detach().to(device, torch.float32).contiguous()


def mconv2d(srcs, mw: MConvWeights, stride=1, pad=1, pad_mode="zeros", act="none", slope=0.0, residual=None, clamp=None,
            out_f32=False, pool=False):
    """16-bit Conv2d over the channel concat of ``srcs`` on the matrix cores.  ``pool=True`` also returns ``F.max_pool2d(out, 2)``
    (written by the same launch for the narrow ReLU layers): ``(out, pooled)``."""
    if isinstance(srcs, torch.Tensor):
        srcs = [srcs]
    for s in srcs:
        _chk16(s)
        if s.dtype != mw.dtype:
            raise capi.IEError("mconv2d: source dtype differs from the packed weights")
    cs = [int(s.shape[1]) for s in srcs]
    if cs != mw.split:
        raise capi.IEError(f"mconv2d: sources carry channels {cs}, weights were packed for {mw.split}")
    n = srcs[0].shape[0]
    # sources smaller than the largest one are read through a nearest-neighbour up-sampling to its size, folded into the kernel's
    # addressing (F.interpolate(x, size=...) in front of the conv: retinexnet/model.py:79-86)
    h, w = max(int(s.shape[2]) for s in srcs), max(int(s.shape[3]) for s in srcs)
    odt = torch.float32 if out_f32 else mw.dtype
    sh = int(getattr(mw, "shuffle", 0))
    d = capi.MConvDesc(n, h, w, len(srcs), mw.cout, mw.kh, mw.kw, stride, pad, capi.PAD[pad_mode], capi.ACT[act], slope,
                       0 if clamp is None else 1, 0.0 if clamp is None else clamp[0], 0.0 if clamp is None else clamp[1],
                       IE_DT[mw.dtype], IE_DT[odt], 0, 0)
    d.pixel_shuffle = sh
    for i, s_ in enumerate(srcs):
        if (int(s_.shape[2]), int(s_.shape[3])) != (h, w):
            d.src_h[i], d.src_w[i] = int(s_.shape[2]), int(s_.shape[3])
    ho = (h + 2 * pad - mw.kh) // stride + 1
    wo = (w + 2 * pad - mw.kw) // stride + 1
    if sh:                                # conv + PixelShuffle(sh) in one kernel: (N, Cout / 4, 2 Ho, 2 Wo)
        out = torch.empty((n, mw.cout // (sh * sh), ho * sh, wo * sh), device=srcs[0].device, dtype=odt, memory_format=CL)
    else:
        out = torch.empty((n, mw.cout, ho, wo), device=srcs[0].device, dtype=odt, memory_format=CL)
    _count_macs(n, ho, wo, mw.cout, sum(cs), mw.kh, mw.kw)
    if residual is not None:
        _chk16(residual)
        if tuple(residual.shape) != tuple(out.shape):
            raise capi.IEError("mconv2d: residual shape mismatch")
    ptrs = (C.c_void_p * len(srcs))(*[s.data_ptr() for s in srcs])
    chans = (C.c_int * len(srcs))(*cs)
    if pool:
        if residual is not None or clamp is not None or out_f32 or sh or stride != 1:
            raise capi.IEError("mconv2d: pool=True takes a plain stride-1 conv (no residual / clamp / fp32 output / pixel shuffle)")
        pooled = torch.empty((n, mw.cout, ho // 2, wo // 2), device=srcs[0].device, dtype=odt, memory_format=CL)
        with torch.cuda.device(out.device):
            capi.check(capi.lib().ie_mconv_pool_nhwc(C.byref(d), ptrs, chans, capi.ptr(mw.w), capi.ptr(mw.b), capi.ptr(out), capi.ptr(pooled),
                                                     capi.current_stream()), "ie_mconv_pool_nhwc")
        return out, pooled
    with torch.cuda.device(out.device):
        capi.check(capi.lib().ie_mconv_nhwc(C.byref(d), ptrs, chans, capi.ptr(mw.w), capi.ptr(mw.b), capi.ptr(residual), capi.ptr(out),
                                            capi.current_stream()), "ie_mconv_nhwc")
    return out


def mconv2d_first_pool_applicable(x8, mw1: MConvWeights, mw2: MConvWeights) -> bool:
    """Does the one-launch first DoubleConv (+ pool) cover these two layers?  (8-channel image, 3x3 -> C -> C with C 32 / 64, ReLU both)"""
    c = mw1.cout
    if not (mw1.split == [8] and mw2.split == [c] and mw2.cout == c and mw1.kh == mw1.kw == mw2.kh == mw2.kw == 3 and mw1.dtype == mw2.dtype == x8.dtype
            and not getattr(mw1, "shuffle", 0) and not getattr(mw2, "shuffle", 0) and x8.shape[1] == 8):
        return False
    return bool(capi.lib().ie_mconv_first_applicable(c, int(x8.shape[2]), int(x8.shape[3])))


def mconv2d_first_pool(x8, mw1: MConvWeights, mw2: MConvWeights):
    """``y = relu(conv2(relu(conv1(x8))))`` and ``F.max_pool2d(y, 2)`` in one launch (unet.py:43-48, 70-72, 152); the conv1 output never reaches HBM."""
    _chk16(x8)
    if not mconv2d_first_pool_applicable(x8, mw1, mw2):
        raise capi.IEError("mconv2d_first_pool: not covered (8-channel image, 3x3 convs 8 -> C -> C with C 32 / 64, W >= 64)")
    n, _, h, w = x8.shape
    c = mw1.cout
    out = torch.empty((n, c, h, w), device=x8.device, dtype=x8.dtype, memory_format=CL)
    pooled = torch.empty((n, c, h // 2, w // 2), device=x8.device, dtype=x8.dtype, memory_format=CL)
    _count_macs(n, h, w, c, 8, 3, 3)
    _count_macs(n, h, w, c, c, 3, 3)
    with torch.cuda.device(out.device):
        capi.check(capi.lib().ie_mconv_first_pool_nhwc(IE_DT[x8.dtype], n, h, w, c, capi.ptr(x8), capi.ptr(mw1.w), capi.ptr(mw1.b), capi.ptr(mw2.w),
                                                       capi.ptr(mw2.b), capi.ptr(out), capi.ptr(pooled), capi.current_stream()), "ie_mconv_first_pool_nhwc")
    return out, pooled


class MTailWeights:
    """A 1x1 Conv2d (at most 16 output channels) that ie_mconv_tail_nhwc contracts behind a narrow 3x3 ReLU layer: 16-bit [16][Cin]
    (rows >= Cout zero) + fp32 bias [16]."""

    def __init__(self, w_oihw: torch.Tensor, bias, dtype, device):
        w = w_oihw.detach().to("cpu", torch.float32)
        if w.dim() != 4 or w.shape[2] != 1 or w.shape[3] != 1 or w.shape[0] > 16:
            raise capi.IEError("MTailWeights: a 1x1 conv with at most 16 output channels")
        self.cout, self.cin, self.dtype = int(w.shape[0]), int(w.shape[1]), dtype
        wz = torch.zeros(16, self.cin)
        wz[:self.cout] = w[:, :, 0, 0]
        bz = torch.zeros(16)
        if bias is not None:
            bz[:self.cout] = bias.detach().to("cpu", torch.float32)
        self.w, self.b = wz.to(device, dtype).contiguous(), bz.to(device).contiguous()


def _mconv_desc(srcs, mw, pad, pad_mode, act, slope):
    cs = [int(s.shape[1]) for s in srcs]
    n = srcs[0].shape[0]
    h, w = max(int(s.shape[2]) for s in srcs), max(int(s.shape[3]) for s in srcs)
    d = capi.MConvDesc(n, h, w, len(srcs), mw.cout, mw.kh, mw.kw, 1, pad, capi.PAD[pad_mode], capi.ACT[act], slope, 0, 0.0, 0.0,
                       IE_DT[mw.dtype], IE_DT[mw.dtype], 0, 0)
    for i, s_ in enumerate(srcs):
        if (int(s_.shape[2]), int(s_.shape[3])) != (h, w):
            d.src_h[i], d.src_w[i] = int(s_.shape[2]), int(s_.shape[3])
    return d, cs, n, h, w


def mconv2d_tail_applicable(srcs, mw: MConvWeights, tail: MTailWeights, pad=1, pad_mode="zeros", act="relu") -> bool:
    """Does the one-launch form (3x3 ReLU conv -> 1x1 conv -> NCHW fp32) cover this layer pair?"""
    if isinstance(srcs, torch.Tensor):
        srcs = [srcs]
    if getattr(mw, "shuffle", 0) or tail.cin != mw.cout or tail.dtype != mw.dtype:
        return False
    d, cs, *_ = _mconv_desc(srcs, mw, pad, pad_mode, act, 0.0)
    return bool(capi.lib().ie_mconv_tail_applicable(C.byref(d), (C.c_int * len(cs))(*cs), tail.cout))


def mconv2d_tail(srcs, mw: MConvWeights, tail: MTailWeights, pad=1, pad_mode="zeros", act="relu", clamp=None) -> torch.Tensor:
    """``clamp(conv1x1(relu(conv3x3(cat(srcs)))))`` as NCHW fp32 in one launch (the U-Net head: unet.py:186-195); the 3x3 layer's
    output never reaches HBM.  Raises where ``mconv2d_tail_applicable`` says no."""
    if isinstance(srcs, torch.Tensor):
        srcs = [srcs]
    for s in srcs:
        _chk16(s)
        if s.dtype != mw.dtype:
            raise capi.IEError("mconv2d_tail: source dtype differs from the packed weights")
    if tail.cin != mw.cout or tail.dtype != mw.dtype:
        raise capi.IEError("mconv2d_tail: the 1x1 weights do not match the 3x3 layer")
    d, cs, n, h, w = _mconv_desc(srcs, mw, pad, pad_mode, act, 0.0)
    if cs != mw.split:
        raise capi.IEError(f"mconv2d_tail: sources carry channels {cs}, weights were packed for {mw.split}")
    out = torch.empty((n, tail.cout, h, w), device=srcs[0].device, dtype=torch.float32)
    _count_macs(n, h, w, mw.cout, sum(cs), mw.kh, mw.kw)
    _count_macs(n, h, w, tail.cout, mw.cout, 1, 1)
    ptrs = (C.c_void_p * len(srcs))(*[s.data_ptr() for s in srcs])
    chans = (C.c_int * len(srcs))(*cs)
    with torch.cuda.device(out.device):
        capi.check(capi.lib().ie_mconv_tail_nhwc(C.byref(d), ptrs, chans, capi.ptr(mw.w), capi.ptr(mw.b), capi.ptr(tail.w), capi.ptr(tail.b), tail.cout,
                                                 0 if clamp is None else 1, 0.0 if clamp is None else clamp[0], 0.0 if clamp is None else clamp[1],
                                                 capi.ptr(out), capi.current_stream()), "ie_mconv_tail_nhwc")
    return out


def depthwise3x3_16(x, w_tap_c: torch.Tensor, bias, act="none", slope=0.0) -> torch.Tensor:
    """16-bit depthwise 3x3 (zero padding 1): ``w_tap_c`` fp32 device tensor [9, C]."""
    _chk16(x)
    n, c, h, w = x.shape
    y = torch.empty_like(x, memory_format=CL)
    _count_macs(n, h, w, c, 1, 3, 3)
    with torch.cuda.device(x.device):
        capi.check(capi.lib().ie_dwconv3x3_nhwc(capi.ptr(x), capi.ptr(w_tap_c), capi.ptr(bias), capi.ptr(y), _dt(x), n, h, w, c, capi.ACT[act], slope,
                                                capi.current_stream()), "ie_dwconv3x3_nhwc")
    return y


def dsconv2d(srcs, dw_w: torch.Tensor, dw_b, pw: MConvWeights, act="none", slope=0.0, out_f32=False) -> torch.Tensor:
    """Fused depthwise 3x3 + pointwise 1x1 over the channel concat of ``srcs`` (16-bit): ``dw_w`` fp32 device [Cin, 9],
    ``pw`` = MConvWeights of the 1x1 conv packed for the same channel split."""
    if isinstance(srcs, torch.Tensor):
        srcs = [srcs]
    for s in srcs:
        _chk16(s)
    cs = [int(s.shape[1]) for s in srcs]
    if cs != pw.split or pw.kh != 1 or pw.kw != 1:
        raise capi.IEError(f"dsconv2d: sources {cs} do not match the packed 1x1 weights {pw.split}")
    n, _, h, w = srcs[0].shape
    odt = torch.float32 if out_f32 else pw.dtype
    d = capi.MConvDesc(n, h, w, len(srcs), pw.cout, 3, 3, 1, 1, 0, capi.ACT[act], slope, 0, 0.0, 0.0, IE_DT[pw.dtype], IE_DT[odt], 0, 0)
    out = torch.empty((n, pw.cout, h, w), device=srcs[0].device, dtype=odt, memory_format=CL)
    _count_macs(n, h, w, sum(cs), 1, 3, 3)
    _count_macs(n, h, w, pw.cout, sum(cs), 1, 1)
    ptrs = (C.c_void_p * len(srcs))(*[s.data_ptr() for s in srcs])
    chans = (C.c_int * len(srcs))(*cs)
    with torch.cuda.device(out.device):
        capi.check(capi.lib().ie_dsconv_nhwc(C.byref(d), ptrs, chans, capi.ptr(dw_w), capi.ptr(dw_b), capi.ptr(pw.w), capi.ptr(pw.b), capi.ptr(out),
                                             capi.current_stream()), "ie_dsconv_nhwc")
    return out


def mconv2d_pair_applicable(x, mwa: MConvWeights, mwb: MConvWeights, pad_mode="zeros", act_a="relu", act_b="relu") -> bool:
    """Can ``relu(conv_b(relu(conv_a(x))))`` -- two stacked 3x3 / stride 1 / padding 1 layers -- run as ONE launch with the intermediate in LDS only?"""
    if not isinstance(x, torch.Tensor) or x.dim() != 4 or mwa.dtype != mwb.dtype or x.dtype != mwa.dtype:
        return False
    if mwa.split != [int(x.shape[1])] or mwb.split != [mwa.cout] or getattr(mwa, "shuffle", 0) or getattr(mwb, "shuffle", 0):
        return False
    if (mwa.kh, mwa.kw, mwb.kh, mwb.kw) != (3, 3, 3, 3):
        return False
    return bool(capi.lib().ie_mconv_pair_applicable(int(x.shape[1]), mwa.cout, mwb.cout, 3, 3, 1, 1, capi.PAD[pad_mode], capi.ACT[act_a], capi.ACT[act_b],
                                                    int(x.shape[3])))


def mconv2d_pair(x, mwa: MConvWeights, mwb: MConvWeights, pad_mode="zeros"):
    """``relu(conv_b(relu(conv_a(x))))`` in one launch (ie_mconv_pair_nhwc): the same bits as two ``mconv2d(..., act="relu")`` calls, without the
    intermediate tensor's trip through HBM.  Raises where ``mconv2d_pair_applicable`` says no."""
    _chk16(x)
    if not mconv2d_pair_applicable(x, mwa, mwb, pad_mode):
        raise capi.IEError("mconv2d_pair: two 3x3 / stride 1 / padding 1 / ReLU layers 64 -> 64 -> 64 on a 16-bit NHWC tensor with W >= 64")
    n, c, h, w = (int(v) for v in x.shape)
    out = torch.empty((n, mwb.cout, h, w), device=x.device, dtype=x.dtype, memory_format=CL)
    _count_macs(n, h, w, mwa.cout, c, 3, 3)
    _count_macs(n, h, w, mwb.cout, mwa.cout, 3, 3)
    with torch.cuda.device(x.device):
        capi.check(capi.lib().ie_mconv_pair_nhwc(capi.ptr(x), capi.ptr(mwa.w), capi.ptr(mwa.b), capi.ptr(mwb.w), capi.ptr(mwb.b), capi.ptr(out),
                                                 IE_DT[x.dtype], n, h, w, capi.PAD[pad_mode], capi.current_stream()), "ie_mconv_pair_nhwc")
    return out


def mconv2d_pair_pool_applicable(x, mwa: MConvWeights, mwb: MConvWeights, pad_mode="zeros") -> bool:
    """Can ``y = relu(conv_b(relu(conv_a(x))))`` with a 32-channel ``x`` and ``MaxPool2d(2)(y)`` run as ONE launch?"""
    if not isinstance(x, torch.Tensor) or x.dim() != 4 or mwa.dtype != mwb.dtype or x.dtype != mwa.dtype:
        return False
    if mwa.split != [int(x.shape[1])] or mwb.split != [mwa.cout] or getattr(mwa, "shuffle", 0) or getattr(mwb, "shuffle", 0):
        return False
    if (mwa.kh, mwa.kw, mwb.kh, mwb.kw) != (3, 3, 3, 3):
        return False
    return bool(capi.lib().ie_mconv_pair_pool_applicable(int(x.shape[1]), mwa.cout, mwb.cout, 3, 3, 1, 1, capi.PAD[pad_mode], capi.ACT["relu"], capi.ACT["relu"],
                                                         int(x.shape[2]), int(x.shape[3])))


def mconv2d_pair_pool(x, mwa: MConvWeights, mwb: MConvWeights, pad_mode="zeros"):
    """``(y, MaxPool2d(2)(y))`` for ``y = relu(conv_b(relu(conv_a(x))))`` in one launch (ie_mconv_pair_pool_nhwc): a U-Net encoder level whose first
    layer has 32 input channels.  Bit-identical to the two-launch path ``mconv2d`` + ``mconv2d(..., pool=True)`` (same packed weights, same
    contraction order, the intermediate rounded once to the storage type: ``torch.equal`` in test_stacked_conv_32_64_64_with_pooled_output); against an
    fp32 torch convolution of the same 16-bit operands both paths carry the storage rounding of the intermediate (<= 3e-3 fp16 / 2e-2 bf16 there)."""
    _chk16(x)
    if not mconv2d_pair_pool_applicable(x, mwa, mwb, pad_mode):
        raise capi.IEError("mconv2d_pair_pool: 3x3 / stride 1 / padding 1 / ReLU layers 32 -> 64 -> 64 on a 16-bit NHWC tensor with W >= 64, H >= 2")
    n, c, h, w = (int(v) for v in x.shape)
    out = torch.empty((n, mwb.cout, h, w), device=x.device, dtype=x.dtype, memory_format=CL)
    pooled = torch.empty((n, mwb.cout, h // 2, w // 2), device=x.device, dtype=x.dtype, memory_format=CL)
    _count_macs(n, h, w, mwa.cout, c, 3, 3)
    _count_macs(n, h, w, mwb.cout, mwa.cout, 3, 3)
    with torch.cuda.device(x.device):
        capi.check(capi.lib().ie_mconv_pair_pool_nhwc(capi.ptr(x), capi.ptr(mwa.w), capi.ptr(mwa.b), capi.ptr(mwb.w), capi.ptr(mwb.b), capi.ptr(out), capi.ptr(pooled),
                                                      IE_DT[x.dtype], n, h, w, capi.PAD[pad_mode], capi.current_stream()), "ie_mconv_pair_pool_nhwc")
    return out, pooled


def mconv2d_pair_thin_applicable(x, mwa: MConvWeights, mwb: MConvWeights, pad_mode="zeros", act_a="relu", act_b="none") -> bool:
    """Can ``act_b(conv_b(relu(conv_a(x))))`` with a thin conv_b (4 .. 16 output channels, fp32 output) run as ONE launch?"""
    if not isinstance(x, torch.Tensor) or x.dim() != 4 or mwa.dtype != mwb.dtype or x.dtype != mwa.dtype:
        return False
    if mwa.split != [int(x.shape[1])] or mwb.split != [mwa.cout] or getattr(mwa, "shuffle", 0) or getattr(mwb, "shuffle", 0):
        return False
    if (mwa.kh, mwa.kw, mwb.kh, mwb.kw) != (3, 3, 3, 3):
        return False
    return bool(capi.lib().ie_mconv_pair_thin_applicable(int(x.shape[1]), mwa.cout, mwb.cout, 3, 3, 1, 1, capi.PAD[pad_mode], capi.ACT[act_a], capi.ACT[act_b],
                                                         int(x.shape[3])))


def mconv2d_pair_thin(x, mwa: MConvWeights, mwb: MConvWeights, pad_mode="zeros", act_b="none", copy16=False, copy16_channels=8):
    """``act_b(conv_b(relu(conv_a(x))))`` in one launch (ie_mconv_pair_thin_nhwc): conv_b has 4 .. 16 output channels and leaves in fp32.
    ``copy16=True`` (4 output channels only) also returns the output as the compute dtype with ``copy16_channels`` = 4 channels or padded
    to 8 (what the next 16-bit network reads), written by the same launch: ``(out_f32, out16)``."""
    _chk16(x)
    if not mconv2d_pair_thin_applicable(x, mwa, mwb, pad_mode, "relu", act_b):
        raise capi.IEError("mconv2d_pair_thin: Conv3x3(64,64)+ReLU then Conv3x3(64, 4..16) on a 16-bit NHWC tensor with W >= 64")
    n, c, h, w = (int(v) for v in x.shape)
    out = torch.empty((n, mwb.cout, h, w), device=x.device, dtype=torch.float32, memory_format=CL)
    out16 = torch.empty((n, int(copy16_channels), h, w), device=x.device, dtype=x.dtype, memory_format=CL) if copy16 else None
    _count_macs(n, h, w, mwa.cout, c, 3, 3)
    _count_macs(n, h, w, mwb.cout, mwa.cout, 3, 3)
    with torch.cuda.device(x.device):
        capi.check(capi.lib().ie_mconv_pair_thin_nhwc(capi.ptr(x), capi.ptr(mwa.w), capi.ptr(mwa.b), capi.ptr(mwb.w), capi.ptr(mwb.b), capi.ptr(out),
                                                      capi.ptr(out16) if copy16 else None, int(copy16_channels), IE_DT[x.dtype], n, h, w, mwb.cout, capi.ACT[act_b],
                                                      capi.PAD[pad_mode], capi.current_stream()), "ie_mconv_pair_thin_nhwc")
    return (out, out16) if copy16 else out


class UNetFirstWeights:
    """Fragment image of a U-Net's first conv (32, 3, 3, 3) (BatchNorm folded) for ie_unet_first_pool_nchw (+ fp32 bias)."""

    def __init__(self, w_oihw: torch.Tensor, bias, dtype, device):
        w = w_oihw.detach().to("cpu", torch.float32).contiguous()
        if tuple(w.shape) != (32, 3, 3, 3):
            raise capi.IEError(f"UNetFirstWeights: (32, 3, 3, 3) expected, got {tuple(w.shape)}")
        self.dtype = dtype
        L = capi.lib()
        self.w = torch.empty(int(L.ie_unet_first_packed_bytes()), device=device, dtype=torch.uint8)
        with torch.cuda.device(self.w.device):
            capi.check(L.ie_unet_first_pack_weights(capi.ptr(w), IE_DT[dtype], capi.ptr(self.w), capi.current_stream()), "ie_unet_first_pack_weights")
        self.b = None if bias is None else bias.detach().to(device, torch.float32).contiguous()


def unet_first_pool_applicable(x_nchw, w1: "UNetFirstWeights", mw2: MConvWeights) -> bool:
    if x_nchw.dim() != 4 or x_nchw.shape[1] != 3 or x_nchw.dtype != torch.float32 or not x_nchw.is_cuda or w1.dtype != mw2.dtype:
        return False
    if mw2.split != [32] or mw2.cout != 32 or (mw2.kh, mw2.kw) != (3, 3):
        return False
    return bool(capi.lib().ie_unet_first_applicable(32, int(x_nchw.shape[2]), int(x_nchw.shape[3])))


def unet_first_pool(x_nchw, w1: "UNetFirstWeights", mw2: MConvWeights):
    """fp32 NCHW image -> ``(x1, max_pool2d(x1, 2))`` with x1 = relu(conv2(relu(conv1(image)))) (BatchNorm folded), 16-bit NHWC, in one launch
    (ie_unet_first_pool_nchw)."""
    if not unet_first_pool_applicable(x_nchw, w1, mw2):
        raise capi.IEError("unet_first_pool: (N,3,H,W) fp32 CUDA image, 32 channels, W >= 64")
    x_nchw = x_nchw.contiguous()
    n, _, h, w = (int(v) for v in x_nchw.shape)
    out = torch.empty((n, 32, h, w), device=x_nchw.device, dtype=w1.dtype, memory_format=CL)
    pooled = torch.empty((n, 32, h // 2, w // 2), device=x_nchw.device, dtype=w1.dtype, memory_format=CL)
    _count_macs(n, h, w, 32, 3, 3, 3)
    _count_macs(n, h, w, 32, 32, 3, 3)
    with torch.cuda.device(x_nchw.device):
        capi.check(capi.lib().ie_unet_first_pool_nchw(capi.ptr(x_nchw), capi.ptr(w1.w), capi.ptr(w1.b), capi.ptr(mw2.w), capi.ptr(mw2.b), capi.ptr(out), capi.ptr(pooled),
                                                      IE_DT[w1.dtype], n, h, w, capi.current_stream()), "ie_unet_first_pool_nchw")
    return out, pooled


def unet_up_head_applicable(skip, up, mwa: MConvWeights, mwb: MConvWeights, tail: MTailWeights) -> bool:
    """Can the last decoder level (cat -> DoubleConv -> 1x1 head) run as one launch (ie_unet_up_head_nhwc)?  ``up``: the skip's resolution or half of it."""
    if skip.dim() != 4 or up.dim() != 4 or skip.dtype != up.dtype or skip.dtype != mwa.dtype or mwb.dtype != mwa.dtype or tail.dtype != mwa.dtype:
        return False
    c = int(skip.shape[1])
    if int(up.shape[1]) != c or mwa.split != [c, c] or mwa.cout != c or mwb.split != [c] or mwb.cout != c or tail.cin != c:
        return False
    if (mwa.kh, mwa.kw, mwb.kh, mwb.kw) != (3, 3, 3, 3) or getattr(mwa, "shuffle", 0) or getattr(mwb, "shuffle", 0):
        return False
    return bool(capi.lib().ie_unet_up_head_applicable(c, tail.cout, int(skip.shape[2]), int(skip.shape[3]), int(up.shape[2]), int(up.shape[3])))


def unet_up_head(skip, up, mwa: MConvWeights, mwb: MConvWeights, tail: MTailWeights, clamp=None):
    """``conv1x1(relu(conv_b(relu(conv_a(cat([skip, up]))))))`` (+ clamp) -> NCHW fp32 in one launch; a half-resolution ``up`` is up-sampled x2
    (bilinear, align_corners=True) inside the kernel."""
    _chk16(skip); _chk16(up)
    if not unet_up_head_applicable(skip, up, mwa, mwb, tail):
        raise capi.IEError("unet_up_head: 32-channel 16-bit NHWC sources, W >= 64, `up` at the skip's resolution or exactly half of it")
    n, c, h, w = (int(v) for v in skip.shape)
    out = torch.empty((n, tail.cout, h, w), device=skip.device, dtype=torch.float32)
    _count_macs(n, h, w, c, 2 * c, 3, 3)
    _count_macs(n, h, w, c, c, 3, 3)
    _count_macs(n, h, w, tail.cout, c, 1, 1)
    lo, hi = clamp if clamp is not None else (0.0, 0.0)
    with torch.cuda.device(skip.device):
        capi.check(capi.lib().ie_unet_up_head_nhwc(capi.ptr(skip), capi.ptr(up), int(up.shape[2]), int(up.shape[3]), capi.ptr(mwa.w), capi.ptr(mwa.b),
                                                   capi.ptr(mwb.w), capi.ptr(mwb.b), capi.ptr(tail.w), capi.ptr(tail.b), tail.cout, int(clamp is not None),
                                                   float(lo), float(hi), capi.ptr(out), IE_DT[skip.dtype], n, h, w, capi.current_stream()),
                   "ie_unet_up_head_nhwc")
    return out


def retinex_decom_input(x_nchw: torch.Tensor, dtype, channels: int = 8) -> torch.Tensor:
    """fp32 NCHW RGB -> ``cat(max_c(x), x)`` as 16-bit NHWC with 4 channels or padded to 8, one pass (RetinexNet DecomNet's input, model.py:45-46)."""
    if x_nchw.dim() != 4 or x_nchw.shape[1] != 3 or x_nchw.dtype != torch.float32 or not x_nchw.is_cuda:
        raise ValueError("retinex_decom_input: (N,3,H,W) fp32 CUDA tensor")
    x_nchw = x_nchw.contiguous()
    n, _, h, w = (int(v) for v in x_nchw.shape)
    y = torch.empty((n, int(channels), h, w), device=x_nchw.device, dtype=dtype, memory_format=CL)
    with torch.cuda.device(x_nchw.device):
        capi.check(capi.lib().ie_retinex_decom_input(capi.ptr(x_nchw), capi.ptr(y), IE_DT[dtype], int(channels), n, h, w, capi.current_stream()),
                   "ie_retinex_decom_input")
    return y


class Conv9x9C4Weights:
    """Fragment image of a Conv2d(4, 64, k, padding=k // 2, replicate), k = 9 or 3, for ie_conv9x9_c4_nhwc / ie_conv3x3_c4_nhwc (+ fp32 bias)."""

    def __init__(self, w_oihw: torch.Tensor, bias, dtype, device):
        w = w_oihw.detach().to("cpu", torch.float32).contiguous()
        if tuple(w.shape) not in ((64, 4, 9, 9), (64, 4, 3, 3)):
            raise capi.IEError(f"Conv9x9C4Weights: (64, 4, 9, 9) or (64, 4, 3, 3) expected, got {tuple(w.shape)}")
        self.dtype, self.k = dtype, int(w.shape[2])
        L = capi.lib()
        nbytes, pack = (L.ie_conv9x9_c4_packed_bytes, L.ie_conv9x9_c4_pack_weights) if self.k == 9 else (L.ie_conv3x3_c4_packed_bytes, L.ie_conv3x3_c4_pack_weights)
        self.w = torch.empty(int(nbytes()), device=device, dtype=torch.uint8)
        with torch.cuda.device(self.w.device):
            capi.check(pack(capi.ptr(w), IE_DT[dtype], capi.ptr(self.w), capi.current_stream()), "ie_conv%dx%d_c4_pack_weights" % (self.k, self.k))
        self.b = None if bias is None else bias.detach().to(device, torch.float32).contiguous()


def conv9x9_c4(x4, cw: Conv9x9C4Weights):
    """Conv2d(4, 64, k, padding k // 2 replicate), k = 9 or 3 per the weights, on a 4-channel 16-bit NHWC image (ie_conv9x9_c4_nhwc / ie_conv3x3_c4_nhwc)."""
    _chk16(x4)
    n, c, h, w = (int(v) for v in x4.shape)
    if c != 4 or x4.dtype != cw.dtype:
        raise capi.IEError("conv9x9_c4: a 4-channel NHWC tensor of the weights' dtype")
    out = torch.empty((n, 64, h, w), device=x4.device, dtype=x4.dtype, memory_format=CL)
    _count_macs(n, h, w, 64, 4, cw.k, cw.k)
    fn = capi.lib().ie_conv9x9_c4_nhwc if cw.k == 9 else capi.lib().ie_conv3x3_c4_nhwc
    with torch.cuda.device(x4.device):
        capi.check(fn(capi.ptr(x4), capi.ptr(cw.w), capi.ptr(cw.b), capi.ptr(out), IE_DT[x4.dtype], n, h, w, capi.current_stream()), "ie_conv%dx%d_c4_nhwc" % (cw.k, cw.k))
    return out


def retinex_outputs(rl, i_delta):
    """DecomNet's (R, L) (fp32 NHWC, 4 channels) and I_delta (fp32, 1 channel) -> NCHW fp32 ``(R, L, S = R * I_delta)`` in one pass."""
    _chk(rl); _chk(i_delta)
    n, c, h, w = (int(v) for v in rl.shape)
    if c != 4 or tuple(i_delta.shape) != (n, 1, h, w):
        raise ValueError("retinex_outputs: rl (N,4,H,W) and i_delta (N,1,H,W), fp32 NHWC")
    R = torch.empty((n, 3, h, w), device=rl.device, dtype=torch.float32)
    L = torch.empty((n, 1, h, w), device=rl.device, dtype=torch.float32)
    S = torch.empty((n, 3, h, w), device=rl.device, dtype=torch.float32)
    with torch.cuda.device(rl.device):
        capi.check(capi.lib().ie_retinex_outputs_f32(capi.ptr(rl), capi.ptr(i_delta), capi.ptr(R), capi.ptr(L), capi.ptr(S), n, h, w, capi.current_stream()),
                   "ie_retinex_outputs_f32")
    return R, L, S


class MConvWeightsDev:
    """Packed weights for ie_mconv_nhwc produced ON THE DEVICE from the trainer's HWIO fp32 master weights (``repack()`` after every
    optimiser step).  ``mode="forward"``: the conv over sources ``split``; ``mode="dgrad"``: the data-gradient conv of source
    ``src_index`` (input = dY with ``cout`` channels, output = that source's channels)."""

    def __init__(self, w_hwio_flat: torch.Tensor, bias, split, cout: int, kh: int, kw: int, dtype, mode="forward", src_index=0):
        self._src, self._split, self._cout_fwd, self._mode, self._si = w_hwio_flat, [int(c) for c in split], int(cout), mode, int(src_index)
        self.kh, self.kw, self.dtype = kh, kw, dtype
        L = capi.lib()
        if mode == "forward":
            self.split, self.cout, self.b = list(self._split), int(cout), bias
        else:
            self.split, self.cout, self.b = [int(cout)], self._split[src_index], None
        chans = (C.c_int * len(self.split))(*self.split)
        nbytes = L.ie_mconv_packed_bytes(chans, len(self.split), self.cout, kh, kw)
        if nbytes == 0:
            capi.check(-1, "ie_mconv_packed_bytes")
        self.w = torch.empty(nbytes, device=w_hwio_flat.device, dtype=torch.uint8)
        self.repack()

    def repack(self):
        chans = (C.c_int * len(self._split))(*self._split)
        with torch.cuda.device(self.w.device):
            capi.check(capi.lib().ie_mconv_pack_weights_dev(capi.ptr(self._src), chans, len(self._split), self._cout_fwd, self.kh, self.kw,
                                                            IE_DT[self.dtype], 0 if self._mode == "forward" else 1, self._si, capi.ptr(self.w),
                                                            capi.current_stream()), "ie_mconv_pack_weights_dev")
