"""The other generators of the hot path on the fp32 NHWC building blocks (``ops.py`` -> ``libie_hip.so``):

  zero_dce++_re   ZeroDCEpp_RE   src/mon/vision/enhance/llie/zero_dce/zero_dcepp.py:86-208 (DSConv2d: nn/modules/conv/dsconv.py:106-149)
  llunet++_re     LLUnetpp_RE    src/mon/vision/enhance/llie/llunetpp/llunetpp.py:97-243
  lllinet         LLLINet        src/mon/vision/enhance/llie/lllinet/lllinet.py:97-260 (normalization.py:177-224, attention.py:1156-1182)
  mprnet          MPRNet         src/mon/vision/enhance/multitask/mprnet/mprnet.py:33-548
  unet            UNet           src/mon/vision/segment/unet/unet.py:35-197  (+ ``upsampler="pixelshuffle"``: nn/modules/encoder/edsr.py:88-123)
  zero_ig_enhance Enhance        src/mon/vision/enhance/llie/zero_ig/zero_ig.py:307-348
  zero_ig_re      ZeroIG         src/mon/vision/enhance/llie/zero_ig/zero_ig.py:257-304, 355-452 (predicting branch)
  rrdnet_re       RRDNet_RE      src/mon/vision/enhance/llie/rrdnet/rrdnet.py:183-289
  zsn2n           ZSN2N          src/mon/vision/enhance/denoise/zsn2n/zsn2n.py:40-124
  zero_dce_v      ZeroDCEV       src/mon/vision/enhance/llie/zero_dce/zero_dce_v.py:87-230 (+ filtering/guided_filter.py:150-204)
  gcenet          GCENet         src/mon/vision/enhance/llie/gcenet/gcenet.py:203-309, 338-502 (+ guided_filter.py:99-147, prior/image.py:221-331)
  retinexnet      RetinexNet     src/mon_extra/vision/enhance/llie/retinexnet/model.py:21-90, 199-238

Same registry names, constructor hyper-parameters, checkpoint key names and output dicts as the reference
classes; the arithmetic runs in the library's kernels (no torch ops on the data path; torch only
allocates).  This is the parity path (fp32, VALU conv); the MFMA path exists for Zero-DCE only so far.
"""
from __future__ import annotations

import math
import os
from collections import OrderedDict
from typing import Any

import torch

from . import capi, ops
from .globals_ import Scheme, Task
from .model import ImageEnhancementModel
from .registry import MODELS


class _GraphModel(ImageEnhancementModel):
    """Host fp32 parameters + per-device repacked weights, rebuilt lazily."""

    def __init__(self, *args, **kwargs):
        # engine extra: "f32" = parity path (VALU fp32 kernels), "f16" / "bf16" = matrix-core path (16-bit NHWC activations,
        # fp32 accumulate) for the models that implement it
        cd = str(kwargs.pop("compute_dtype", "f32")).lower()
        if cd not in capi.DTYPES:
            raise ValueError(f"`compute_dtype` must be one of f32/f16/bf16, but got {cd}.")
        self.compute_dtype = {capi.IE_F32: "f32", capi.IE_F16: "f16", capi.IE_BF16: "bf16"}[capi.DTYPES[cd]]
        super().__init__(*args, **kwargs)
        self._dev_cache = None

    @property
    def _tdt(self) -> torch.dtype:
        return ops.TORCH_DT[self.compute_dtype]

    def _input_any(self, datapoint: dict):
        """Image as the compute dtype's NHWC tensor: fp32 with C channels, or 16-bit zero-padded to 8 channels."""
        if self.compute_dtype == "f32":
            return self._input(datapoint)
        self.assert_datapoint(datapoint)
        self._require_cuda()
        x = datapoint.get("image")
        if x.dim() == 3:
            x = x[None]
        if x.dim() != 4 or x.shape[1] != self.in_channels:
            raise ValueError(f"`image` must be (N,{self.in_channels},H,W), but got {tuple(x.shape)}.")
        return ops.to_nhwc16(x.to(self.device), self._tdt, 8)          # NCHW fp32 -> NHWC 16-bit, channels padded, in one pass

    def _wb(self, name, bn=None, negate=False, cout=None):
        """Host (weight, bias) of conv ``name``: eval BatchNorm ``bn`` folded in, optionally only the first ``cout`` output
        channels and/or negated (``x - conv(x)`` then runs as one conv with a fused residual: -(W*x + b) = (-W)*x + (-b) exactly)."""
        w, b = self._params[f"{name}.weight"], self._params.get(f"{name}.bias")
        if bn is not None:
            w, b = ops.fold_bn(w, b, self._params[f"{bn}.weight"], self._params[f"{bn}.bias"],
                               self._params[f"{bn}.running_mean"], self._params[f"{bn}.running_var"])
        if cout is not None:
            w, b = w[:cout], None if b is None else b[:cout]
        if negate:
            w, b = -w, None if b is None else -b
        return w, b

    def _conv(self, srcs, name, bn=None, stride=1, pad=1, pad_mode="zeros", act="none", slope=0.0, residual=None, clamp=None,
              out_f32=False, cout_pad=None, valid=None, negate=False, cout=None, shuffle=0, pool=False):
        """Conv2d ``name`` (+ folded eval BatchNorm ``bn``) over the channel concat of ``srcs`` in the model's compute dtype.
        On the 16-bit path sources may carry zero-padded channels (a 3-channel image stored as 8): the weights of the missing
        input channels are zero; ``cout_pad`` pads the output channels the same way (zero weights and bias)."""
        if isinstance(srcs, torch.Tensor):
            srcs = [srcs]
        if self.compute_dtype == "f32":
            y = ops.conv2d(srcs, self._cw(name, bn, negate, cout), stride=stride, pad=pad, pad_mode=pad_mode, act=act, slope=slope,
                           residual=residual, clamp=clamp)
            if pool:                      # ``pool=True``: (y, nn.MaxPool2d(2)(y)) -- one launch on the 16-bit path where the strip kernel covers the layer
                return y, ops.maxpool2x2(y)
            return ops.pixel_shuffle(y, shuffle) if shuffle else y
        split = tuple(int(t.shape[1]) for t in srcs)
        w_ = self._params[f"{name}.weight"]
        fuse = bool(shuffle) and shuffle == 2 and w_.shape[2] == 3 and w_.shape[0] % 64 == 0 and stride == 1 and residual is None and not out_f32
        key = ("mconv", name, bn, split, cout_pad, None if valid is None else tuple(valid), negate, cout, fuse)
        if self._dev_cache is None:
            self._dev_cache = {}
        if key not in self._dev_cache:
            w, b = self._wb(name, bn, negate, cout)
            cin_map = None
            if sum(split) != w.shape[1]:
                real = list(valid) if valid is not None else ([w.shape[1]] if len(split) == 1 else None)
                if real is None or len(real) != len(split) or sum(real) != w.shape[1] or any(r > c for r, c in zip(real, split)):
                    raise capi.IEError(f"{name}: sources carry {split} channels, weights expect {w.shape[1]}")
                cin_map, base = [], 0
                for r, c in zip(real, split):
                    cin_map += list(range(base, base + r)) + [-1] * (c - r)
                    base += r
            if cout_pad is not None and cout_pad > w.shape[0]:
                extra = cout_pad - w.shape[0]
                w = torch.cat([w, torch.zeros(extra, *w.shape[1:])], 0)
                b = None if b is None else torch.cat([b, torch.zeros(extra)], 0)
            self._dev_cache[key] = ops.MConvWeights(w, b, split, self._tdt, self.device, cin_map, shuffle=2 if fuse else 0)
        y = ops.mconv2d(srcs, self._dev_cache[key], stride=stride, pad=pad, pad_mode=pad_mode, act=act, slope=slope,
                        residual=residual, clamp=clamp, out_f32=out_f32, pool=pool)
        return ops.pixel_shuffle(y, shuffle) if (shuffle and not fuse) else y

    def _mw_plain(self, split, name, bn=None):
        """The cached ie_mconv_nhwc weights of conv ``name`` (+ folded ``bn``) for sources of ``split`` channels -- the entry ``_conv`` builds
        for a plain call (an image source carried as 8 channels gets zero weights for the padding channels)."""
        split = tuple(int(c) for c in split)
        key = ("mconv", name, bn, split, None, None, False, None, False)
        if self._dev_cache is None:
            self._dev_cache = {}
        if key not in self._dev_cache:
            w, b = self._wb(name, bn)
            cin_map = None
            if sum(split) != w.shape[1]:
                if len(split) != 1 or w.shape[1] > split[0]:
                    raise capi.IEError(f"{name}: sources carry {split} channels, weights expect {w.shape[1]}")
                cin_map = list(range(w.shape[1])) + [-1] * (split[0] - w.shape[1])
            self._dev_cache[key] = ops.MConvWeights(w, b, split, self._tdt, self.device, cin_map)
        return self._dev_cache[key]

    def _conv_tail(self, srcs, name, bn, tail_name, clamp=None):
        """``clamp(conv1x1 tail_name (relu(conv3x3 name + folded bn)))`` as NCHW fp32 in ONE launch on the 16-bit path (the 3x3 layer's output
        never reaches HBM), or None where the one-launch form does not cover the pair (fp32 path, wide layers, W % 4 != 0)."""
        if self.compute_dtype == "f32":
            return None
        if isinstance(srcs, torch.Tensor):
            srcs = [srcs]
        split = tuple(int(t.shape[1]) for t in srcs)
        if self._dev_cache is None:
            self._dev_cache = {}
        wshape = self._params[f"{name}.weight"].shape
        if sum(split) != wshape[1] or wshape[2] != 3 or self._params[f"{tail_name}.weight"].shape[0] > 16:     # e.g. a 34-class segmentation head
            return None
        key = ("mconv", name, bn, split, None, None, False, None, False)
        if key not in self._dev_cache:
            self._dev_cache[key] = ops.MConvWeights(*self._wb(name, bn), split, self._tdt, self.device)
        tkey = ("mtail", tail_name)
        if tkey not in self._dev_cache:
            self._dev_cache[tkey] = ops.MTailWeights(self._params[f"{tail_name}.weight"], self._params.get(f"{tail_name}.bias"), self._tdt, self.device)
        mw, tw = self._dev_cache[key], self._dev_cache[tkey]
        if not ops.mconv2d_tail_applicable(srcs, mw, tw):
            return None
        return ops.mconv2d_tail(srcs, mw, tw, clamp=clamp)

    def _ds_block(self, prefix, srcs, act, out_f32=False, slope=0.0):
        """DepthwiseSeparableConv2d ``prefix`` (dsconv.py:106-149: ``prefix.dw_conv`` 3x3 depthwise + ``prefix.pw_conv`` 1x1) over the channel
        concat of ``srcs``, + activation.  Depthwise over the concat == depthwise per source with the matching weight slice."""
        outs, off, valid = [], 0, []
        f32 = self.compute_dtype == "f32"
        if self._dev_cache is None:
            self._dev_cache = {}
        wdw, bdw = self._params[f"{prefix}.dw_conv.weight"], self._params[f"{prefix}.dw_conv.bias"]
        split = tuple(int(t.shape[1]) for t in srcs)
        if not f32 and sum(split) == wdw.shape[0] and split != (8,):
            # fused depthwise + pointwise kernel (the depthwise result stays in LDS); the 8-channel RGB input takes the unfused path
            key = ("dsfused", prefix, split)
            if key not in self._dev_cache:
                pw = ops.MConvWeights(self._params[f"{prefix}.pw_conv.weight"], self._params[f"{prefix}.pw_conv.bias"], list(split), self._tdt, self.device)
                self._dev_cache[key] = (wdw[:, 0].reshape(wdw.shape[0], 9).contiguous().to(self.device), bdw.to(self.device).contiguous(), pw)
            dw_w, dw_b, pw = self._dev_cache[key]
            return ops.dsconv2d(list(srcs), dw_w, dw_b, pw, act=act, slope=slope, out_f32=out_f32)
        for t in srcs:
            c = t.shape[1]
            real = min(c, wdw.shape[0] - off)         # 16-bit path: the RGB input is carried as 8 channels
            key = ("dw", prefix, off, c)
            if key not in self._dev_cache:
                if f32:
                    self._dev_cache[key] = ops.ConvWeights(wdw[off:off + c], bdw[off:off + c], self.device)
                else:
                    wt = torch.zeros(9, c)
                    wt[:, :real] = wdw[off:off + real, 0].reshape(real, 9).t()
                    bt = torch.zeros(c)
                    bt[:real] = bdw[off:off + real]
                    self._dev_cache[key] = (wt.contiguous().to(self.device), bt.to(self.device))
            if f32:
                outs.append(ops.depthwise_conv2d(t, self._dev_cache[key]))
            else:
                outs.append(ops.depthwise3x3_16(t, *self._dev_cache[key]))
            valid.append(real)
            off += real
        return self._conv(outs, f"{prefix}.pw_conv", pad=0, act=act, slope=slope, out_f32=out_f32,
                          valid=None if f32 or valid == [t.shape[1] for t in outs] else valid)

    def _on_params_change(self):
        self._dev_cache = None

    def _on_device_change(self):
        self._dev_cache = None

    def _require_cuda(self):
        if self.device.type != "cuda":
            raise capi.IEError(f"{self.__class__.__name__} runs on the HIP device only (model.device = {self.device}); "
                               f"call .to('cuda').  There is no CPU execution path in this engine.")

    # torch.nn.Conv2d default init: kaiming_uniform(a=sqrt(5)) == U(+-1/sqrt(fan_in)) for weight and bias
    def _init_conv(self, name: str, g: torch.Generator, std: float | None = None, bias: bool = True, zero_bias=False):
        w = self._params[f"{name}.weight"]
        fan_in = w.shape[1] * w.shape[2] * w.shape[3]
        bound = 1.0 / math.sqrt(fan_in)
        if std is None:
            self._params[f"{name}.weight"] = (torch.rand(w.shape, generator=g) * 2 - 1) * bound
        else:
            self._params[f"{name}.weight"] = torch.randn(w.shape, generator=g) * std
        if bias and f"{name}.bias" in self._params:
            b = self._params[f"{name}.bias"]
            self._params[f"{name}.bias"] = torch.zeros_like(b) if zero_bias else (torch.rand(b.shape, generator=g) * 2 - 1) * bound

    def _gen(self, seed):
        g = torch.Generator()
        g.manual_seed(int(seed)) if seed is not None else g.seed()
        return g

    def _cw(self, name: str, bn: str | None = None, negate: bool = False, cout: int | None = None) -> ops.ConvWeights:
        """ConvWeights of conv ``name`` (optionally with eval-mode BatchNorm ``bn`` folded in), cached per device."""
        if self._dev_cache is None:
            self._dev_cache = {}
        key = (name, bn, negate, cout)
        if key not in self._dev_cache:
            self._dev_cache[key] = ops.ConvWeights(*self._wb(name, bn, negate, cout), self.device)
        return self._dev_cache[key]

    def _vec(self, name: str) -> torch.Tensor:
        if self._dev_cache is None:
            self._dev_cache = {}
        if name not in self._dev_cache:
            self._dev_cache[name] = self._params[name].to(self.device, torch.float32).contiguous()
        return self._dev_cache[name]

    def _vec_pad(self, name: str, n: int) -> torch.Tensor:
        """Per-channel vector zero-padded to ``n`` entries (norm affine of zero-padded channels)."""
        if self._dev_cache is None:
            self._dev_cache = {}
        key = ("vecpad", name, n)
        if key not in self._dev_cache:
            v = self._params[name].to(torch.float32)
            self._dev_cache[key] = torch.cat([v, torch.zeros(n - v.numel())]).to(self.device).contiguous()
        return self._dev_cache[key]

    def _conv_transpose2x2(self, x, name):
        """nn.ConvTranspose2d(k=2, s=2).  16-bit path: a 1x1 matrix-core conv to 4*Cout channels (o*4 + i*2 + j) + pixel shuffle."""
        if self.compute_dtype == "f32":
            if self._dev_cache is None:
                self._dev_cache = {}
            if ("ct", name) not in self._dev_cache:
                self._dev_cache[("ct", name)] = ops.ConvT2x2Weights(self._params[f"{name}.weight"], self._params[f"{name}.bias"], self.device)
            return ops.conv_transpose2x2(x, self._dev_cache[("ct", name)])
        key = ("mconvT", name)
        if self._dev_cache is None:
            self._dev_cache = {}
        if key not in self._dev_cache:
            w = self._params[f"{name}.weight"]                            # (Cin, Cout, 2, 2)
            b = self._params.get(f"{name}.bias")
            ci, co = w.shape[0], w.shape[1]
            w1 = w.permute(1, 2, 3, 0).reshape(co * 4, ci, 1, 1).contiguous()
            b1 = None if b is None else b.repeat_interleave(4)
            self._dev_cache[key] = ops.MConvWeights(w1, b1, [ci], self._tdt, self.device)
        return ops.pixel_shuffle(ops.mconv2d(x, self._dev_cache[key], pad=0), 2)

    def _input(self, datapoint: dict) -> torch.Tensor:
        self.assert_datapoint(datapoint)
        self._require_cuda()
        x = datapoint.get("image")
        if x.dim() == 3:
            x = x[None]
        if x.dim() != 4 or x.shape[1] != self.in_channels:
            raise ValueError(f"`image` must be (N,{self.in_channels},H,W), but got {tuple(x.shape)}.")
        return ops.to_nhwc(x.to(self.device, torch.float32))

    def _finish_init(self, seed):
        has_tensor_weights = isinstance(self.weights, dict) and any(k.endswith(".weight") for k in self.weights)
        if self.weights and (not isinstance(self.weights, dict) or "path" in self.weights or has_tensor_weights):
            self.load_weights()
        else:
            self.init_weights(seed)


# ----------------------------------------------------------------------------------------------------------------
@MODELS.register(name="zero_dce++_re", arch="zero_dce")
class ZeroDCEpp_RE(_GraphModel):
    """zero_dcepp.py:86-208.  DSConv2d = depthwise 3x3 (bias) then pointwise 1x1 (bias), state_dict keys
    ``e_conv{k}.dw_conv.*`` / ``e_conv{k}.pw_conv.*`` (dsconv.py:122-143); the original repo's
    ``depth_conv`` / ``point_conv`` spellings (zero_dce++/model.py:10-25) are accepted too."""

    arch = "zero_dce"
    tasks = [Task.LLIE]
    schemes = [Scheme.UNSUPERVISED, Scheme.ZERO_REFERENCE]

    def __init__(self, in_channels: int = 3, num_channels: int = 32, num_iters: int = 8, scale_factor: float = 1.0,
                 weights: Any = None, seed: int = None, *args, **kwargs):
        kwargs.pop("name", None)
        super().__init__(name="zero_dce++_re", in_channels=in_channels, weights=weights, *args, **kwargs)
        if isinstance(self.weights, dict):
            in_channels = self.weights.get("in_channels", in_channels)
            num_channels = self.weights.get("num_channels", num_channels)
            num_iters = self.weights.get("num_iters", num_iters)
            scale_factor = self.weights.get("scale_factor", scale_factor)
        self.in_channels, self.num_channels, self.num_iters, self.scale_factor = in_channels, num_channels, num_iters, scale_factor
        c = num_channels
        for k, (ci, co) in enumerate([(in_channels, c), (c, c), (c, c), (c, c), (2 * c, c), (2 * c, c), (2 * c, self.out_channels)], 1):
            self._params[f"e_conv{k}.dw_conv.weight"] = torch.zeros(ci, 1, 3, 3)
            self._params[f"e_conv{k}.dw_conv.bias"] = torch.zeros(ci)
            self._params[f"e_conv{k}.pw_conv.weight"] = torch.zeros(co, ci, 1, 1)
            self._params[f"e_conv{k}.pw_conv.bias"] = torch.zeros(co)
        self._finish_init(seed)

    def map_key(self, k: str) -> str:
        return k.replace(".depth_conv.", ".dw_conv.").replace(".point_conv.", ".pw_conv.")

    def init_weights(self, seed: int = None):
        g = self._gen(seed)
        for k in range(1, 8):          # zero_dcepp.py:152-163: conv weights ~ N(0, 0.02)
            self._init_conv(f"e_conv{k}.dw_conv", g, std=0.02)
            self._init_conv(f"e_conv{k}.pw_conv", g, std=0.02)
        self._on_params_change()

    def _ds(self, k, srcs, act, out_f32=False):
        return self._ds_block(f"e_conv{k}", srcs, act, out_f32)

    def forward(self, datapoint: dict, *args, **kwargs) -> dict:
        x = self._input(datapoint)
        xd = x
        if self.scale_factor != 1:
            h, w = x.shape[2], x.shape[3]
            # F.interpolate(scale_factor=1/s, mode="bilinear"): out = floor(in * (1/s)), align_corners=False
            xd = ops.resize(x, (int(math.floor(h * (1 / self.scale_factor))), int(math.floor(w * (1 / self.scale_factor)))), "bilinear")
        if self.compute_dtype != "f32":
            xd = ops.cast(xd, self._tdt, 8)
        x1 = self._ds(1, [xd], "relu")
        x2 = self._ds(2, [x1], "relu")
        x3 = self._ds(3, [x2], "relu")
        x4 = self._ds(4, [x3], "relu")
        x5 = self._ds(5, [x3, x4], "relu")
        x6 = self._ds(6, [x2, x5], "relu")
        xr = self._ds(7, [x1, x6], "tanh", out_f32=True)          # 3-channel map, fp32 on every path
        if self.scale_factor != 1:
            # nn.UpsamplingBilinear2d(scale_factor): align_corners=True, out = floor(in * s)
            xr = ops.resize(xr, (int(math.floor(xr.shape[2] * self.scale_factor)), int(math.floor(xr.shape[3] * self.scale_factor))),
                            "bilinear_align")
        n, _, h, w = x.shape
        # the curve with one shared 3-channel map for all iterations (zero_dcepp.py:201-203)
        y = torch.empty((n, 3, h, w), device=x.device, dtype=torch.float32)
        xin = ops.to_nchw(x)
        with torch.cuda.device(x.device):
            capi.check(capi.lib().ie_le_curve(capi.ptr(xin), capi.ptr(xr), 3, capi.ptr(y), n, h, w, self.num_iters,
                                              capi.current_stream()), "ie_le_curve")
        return OrderedDict(adjust=ops.to_nchw(xr), enhanced=y)


# ----------------------------------------------------------------------------------------------------------------
@MODELS.register(name="llunet++_re", arch="llunet++")
class LLUnetpp_RE(_GraphModel):
    """llunetpp.py:144-243; block llunetpp.py:97-137.  Checkpoint keys per block ``conv1, norm1, conv2, conv3,
    conv1_2, conv2_3`` (mon) or ``conv_1, norm1, conv_2, conv_3, conv_1_1_1, conv_1_1`` (original,
    llunet++/model.py:9-20); ``loss.*`` ignored."""

    arch = "llunet++"
    tasks = [Task.LLIE]
    schemes = [Scheme.SUPERVISED]
    _BLOCKS = ["conv0_0", "conv1_0", "conv2_0", "conv3_0", "conv4_0", "conv0_1", "conv1_1", "conv2_1", "conv3_1",
               "conv0_2", "conv1_2", "conv2_2", "conv0_3", "conv1_3", "conv0_4"]

    def __init__(self, in_channels: int = 3, weights: Any = None, loss_weights=None, seed: int = None, *args, **kwargs):
        kwargs.pop("name", None)
        super().__init__(name="llunet++_re", in_channels=in_channels, weights=weights, *args, **kwargs)
        if isinstance(self.weights, dict):
            in_channels = self.weights.get("in_channels", in_channels)
        self.in_channels = in_channels
        self.predicting = True     # inference engine: no ref_image needed
        f = [32, 64, 128, 256, 512]
        io = {"conv0_0": (in_channels, f[0]), "conv1_0": (f[0], f[1]), "conv2_0": (f[1], f[2]), "conv3_0": (f[2], f[3]),
              "conv4_0": (f[3], f[4]), "conv0_1": (f[0] + f[1], f[0]), "conv1_1": (f[1] + f[2], f[1]), "conv2_1": (f[2] + f[3], f[2]),
              "conv3_1": (f[3] + f[4], f[3]), "conv0_2": (f[0] * 2 + f[1], f[0]), "conv1_2": (f[1] * 2 + f[2], f[1]),
              "conv2_2": (f[2] * 2 + f[3], f[2]), "conv0_3": (f[0] * 3 + f[1], f[0]), "conv1_3": (f[1] * 3 + f[2], f[1]),
              "conv0_4": (f[0] * 4 + f[1], f[0])}
        self._io = io
        for b in self._BLOCKS:
            ci, co = io[b]
            P = self._params
            P[f"{b}.conv1.weight"], P[f"{b}.conv1.bias"] = torch.zeros(ci, ci, 3, 3), torch.zeros(ci)
            P[f"{b}.norm1.weight"], P[f"{b}.norm1.bias"] = torch.ones(ci), torch.zeros(ci)
            P[f"{b}.conv2.weight"], P[f"{b}.conv2.bias"] = torch.zeros(co, 2 * ci, 3, 3), torch.zeros(co)
            P[f"{b}.conv3.weight"], P[f"{b}.conv3.bias"] = torch.zeros(co, co, 3, 3), torch.zeros(co)
            P[f"{b}.conv1_2.weight"], P[f"{b}.conv1_2.bias"] = torch.zeros(ci, ci, 1, 1), torch.zeros(ci)
            P[f"{b}.conv2_3.weight"], P[f"{b}.conv2_3.bias"] = torch.zeros(co, 2 * ci, 1, 1), torch.zeros(co)
        self._params["final.weight"], self._params["final.bias"] = torch.zeros(self.out_channels, f[0], 1, 1), torch.zeros(self.out_channels)
        self._finish_init(seed)

    def map_key(self, k: str) -> str:
        for a, b in ((".conv_1_1_1.", ".conv1_2."), (".conv_1_1.", ".conv2_3."), (".conv_1.", ".conv1."), (".conv_2.", ".conv2."),
                     (".conv_3.", ".conv3.")):
            k = k.replace(a, b)
        return k

    def init_weights(self, seed: int = None):
        g = self._gen(seed)
        for b in self._BLOCKS:
            for c in ("conv1", "conv2", "conv3", "conv1_2", "conv2_3"):
                self._init_conv(f"{b}.{c}", g)
        self._init_conv("final", g)
        self._on_params_change()

    def _block(self, b: str, srcs):
        c, ci = self._conv, self._io[b][0]
        f32 = self.compute_dtype == "f32"
        cp = None if f32 or ci % 8 == 0 else (ci + 7) // 8 * 8        # conv0_0: 3 channels carried as 8 on the 16-bit path
        x12 = c(srcs, f"{b}.conv1_2", pad=0, cout_pad=cp)
        x1 = c(srcs, f"{b}.conv1", cout_pad=cp)
        if cp is None:
            x1 = ops.instance_norm(x1, self._vec(f"{b}.norm1.weight"), self._vec(f"{b}.norm1.bias"), 1e-5, "lrelu", 0.2)
        else:
            x1 = ops.instance_norm(x1, self._vec_pad(f"{b}.norm1.weight", cp), self._vec_pad(f"{b}.norm1.bias", cp), 1e-5, "lrelu", 0.2)
        v = None if cp is None else [ci, ci]
        x23 = c([x1, x12], f"{b}.conv2_3", pad=0, valid=v)
        x2 = c([x1, x12], f"{b}.conv2", act="lrelu", slope=0.2, valid=v)
        return c(x2, f"{b}.conv3", act="lrelu", slope=0.2, residual=x23)

    supports_mixed_training = True

    def forward_train(self, x, store, tape):
        """Training graph of llunetpp.py:97-137, 218-243 on the fp32 taped ops (instance statistics are per sample in training
        and in inference alike).  Returns the clamped NHWC prediction; the reference's own loss needs a VGG-19 that cannot be
        fetched offline, so the trainer pairs it with L1 / SSIM."""
        from . import train as T
        if x.shape[2] % 16 or x.shape[3] % 16:
            raise ValueError(f"LLUnet++ needs H and W divisible by 16 (got {x.shape[2]}x{x.shape[3]}).")
        g = store.grads
        mixed = store.compute_dtype != "f32"
        if mixed:      # 16-bit graph: RGB carried as 16 channels, the 3-channel head computed as 16; clamp and loss in fp32
            x = T.t_cast(tape, x, ops.TORCH_DT[store.compute_dtype], channels=16)

        def conv(name, srcs, pad=1, act="none", cout_pad=None):
            return T.t_conv(tape, srcs, store.conv(name), *store.conv_grads(name), pad=pad, act=act, slope=0.2, cout_pad=cout_pad)

        def block(b, srcs):
            ci = self._io[b][0]
            cp = 16 if (mixed and ci % 8) else None        # conv0_0 works on the 3 RGB channels: carried as 16 in the 16-bit graph
            x12 = conv(f"{b}.conv1_2", srcs, pad=0, cout_pad=cp)
            x1 = conv(f"{b}.conv1", srcs, cout_pad=cp)
            x1 = T.t_norm(tape, x1, store.view(f"{b}.norm1.weight"), store.view(f"{b}.norm1.bias"), store.view(f"{b}.norm1.weight", g),
                          store.view(f"{b}.norm1.bias", g), per_sample=True, eps=1e-5, act="lrelu", slope=0.2)
            x23 = conv(f"{b}.conv2_3", [x1, x12], pad=0)
            x2 = conv(f"{b}.conv2", [x1, x12], act="lrelu")
            return T.t_add(tape, conv(f"{b}.conv3", [x2], act="lrelu"), x23)

        pool = lambda t: T.t_maxpool(tape, t)
        up = lambda t: T.t_resize(tape, t, (t.shape[2] * 2, t.shape[3] * 2), "bilinear_align")
        x0_0 = block("conv0_0", [x])
        x1_0 = block("conv1_0", [pool(x0_0)])
        x0_1 = block("conv0_1", [x0_0, up(x1_0)])
        x2_0 = block("conv2_0", [pool(x1_0)])
        x1_1 = block("conv1_1", [x1_0, up(x2_0)])
        x0_2 = block("conv0_2", [x0_0, x0_1, up(x1_1)])
        x3_0 = block("conv3_0", [pool(x2_0)])
        x2_1 = block("conv2_1", [x2_0, up(x3_0)])
        x1_2 = block("conv1_2", [x1_0, x1_1, up(x2_1)])
        x0_3 = block("conv0_3", [x0_0, x0_1, x0_2, up(x1_2)])
        x4_0 = block("conv4_0", [pool(x3_0)])
        x3_1 = block("conv3_1", [x3_0, up(x4_0)])
        x2_2 = block("conv2_2", [x2_0, x2_1, up(x3_1)])
        x1_3 = block("conv1_3", [x1_0, x1_1, x1_2, up(x2_2)])
        x0_4 = block("conv0_4", [x0_0, x0_1, x0_2, x0_3, up(x1_3)])
        if mixed:
            y = T.t_cast(tape, conv("final", [x0_4], pad=0, cout_pad=16), torch.float32, channels=self.out_channels)
        else:
            y = conv("final", [x0_4], pad=0)
        return T.t_clamp(tape, y, 0.0, 1.0)

    def forward(self, datapoint: dict, *args, **kwargs) -> dict:
        x = self._input_any(datapoint)
        if x.shape[2] % 16 or x.shape[3] % 16:
            raise ValueError(f"LLUnet++ needs H and W divisible by 16 (got {x.shape[2]}x{x.shape[3]}); use infer(), which "
                             f"resizes to a multiple of 32 like the reference (vision/model.py:106).")
        B, up, pool = self._block, ops.upsample2x_bilinear, ops.maxpool2x2
        x0_0 = B("conv0_0", [x])
        x1_0 = B("conv1_0", [pool(x0_0)])
        x0_1 = B("conv0_1", [x0_0, up(x1_0)])
        x2_0 = B("conv2_0", [pool(x1_0)])
        x1_1 = B("conv1_1", [x1_0, up(x2_0)])
        x0_2 = B("conv0_2", [x0_0, x0_1, up(x1_1)])
        x3_0 = B("conv3_0", [pool(x2_0)])
        x2_1 = B("conv2_1", [x2_0, up(x3_0)])
        x1_2 = B("conv1_2", [x1_0, x1_1, up(x2_1)])
        x0_3 = B("conv0_3", [x0_0, x0_1, x0_2, up(x1_2)])
        x4_0 = B("conv4_0", [pool(x3_0)])
        x3_1 = B("conv3_1", [x3_0, up(x4_0)])
        x2_2 = B("conv2_2", [x2_0, x2_1, up(x3_1)])
        x1_3 = B("conv1_3", [x1_0, x1_1, x1_2, up(x2_2)])
        x0_4 = B("conv0_4", [x0_0, x0_1, x0_2, x0_3, up(x1_3)])
        y = self._conv(x0_4, "final", pad=0, clamp=(0.0, 1.0), out_f32=True)
        return OrderedDict(enhanced=ops.to_nchw(y))


# ----------------------------------------------------------------------------------------------------------------
@MODELS.register(name="lllinet", arch="lllinet")
class LLLINet(_GraphModel):
    """LLLINet (lllinet.py:97-260): the UNet++ grid of LLUnet++ with a different block (lllinet.py:97-156) --
    x1 = SimAM(LeakyReLU(LearnableInstanceNorm(Conv3x3(x)))) (no norm in ``conv0_0``), x2 = Conv3x3(x1), x3 = cat[x2, Conv1x1(x)],
    out = LeakyReLU(Conv3x3(LeakyReLU(Conv3x3(x3)))) + Conv1x1(x3) -- and denser decoder wiring (:235-253), final Conv1x1 + clamp[0, 1].
    Keys per block ``conv1, norm1.{weight,bias,r}, conv2, conv3, conv4, conv1_3, conv3_4``; ``loss.*`` ignored.  fp32 parity path and 16-bit matrix-core path."""

    arch = "lllinet"
    tasks = [Task.LLIE]
    schemes = [Scheme.SUPERVISED]

    def __init__(self, in_channels: int = 3, out_channels: int = 3, weights: Any = None, loss_weights=None, seed: int = None, filters=None, *args, **kwargs):
        kwargs.pop("name", None)
        super().__init__(name="lllinet", in_channels=in_channels, weights=weights, *args, **kwargs)
        if isinstance(self.weights, dict):
            in_channels = self.weights.get("in_channels", in_channels)
            out_channels = self.weights.get("out_channels", out_channels)
        self.in_channels, self._out = in_channels, out_channels
        self.predicting = True
        f = list(filters) if filters else [32, 64, 128, 256, 512]      # (lllinet.py:194 hard-codes the widths; ``filters`` lets a test build the same grid narrow)
        if len(f) != 5:
            raise ValueError(f"LLLINet: filters must name five widths (got {f})")
        io = {"conv0_0": (in_channels, f[0]), "conv1_0": (f[0], f[1]), "conv2_0": (f[1], f[2]), "conv3_0": (f[2], f[3]), "conv4_0": (f[3], f[4]),
              "conv0_1": (f[0] + 2 * f[1], f[0]), "conv1_1": (f[1] + 2 * f[2], f[1]), "conv2_1": (f[2] + 2 * f[3], f[2]), "conv3_1": (f[3] + f[4], f[3]),
              "conv0_2": (2 * f[0] + 2 * f[1], f[0]), "conv1_2": (2 * f[1] + 2 * f[2], f[1]), "conv2_2": (2 * f[2] + f[3], f[2]),
              "conv0_3": (3 * f[0] + 2 * f[1], f[0]), "conv1_3": (3 * f[1] + f[2], f[1]), "conv0_4": (4 * f[0] + f[1], f[0])}
        self._io = io
        P = self._params
        for b, (ci, co) in io.items():
            P[f"{b}.conv1.weight"], P[f"{b}.conv1.bias"] = torch.zeros(ci, ci, 3, 3), torch.zeros(ci)
            if b != "conv0_0":                                           # use_in=False there (lllinet.py:203)
                P[f"{b}.norm1.weight"], P[f"{b}.norm1.bias"], P[f"{b}.norm1.r"] = torch.ones(ci), torch.zeros(ci), torch.full((ci,), 0.5)
            P[f"{b}.conv2.weight"], P[f"{b}.conv2.bias"] = torch.zeros(ci, ci, 3, 3), torch.zeros(ci)
            P[f"{b}.conv3.weight"], P[f"{b}.conv3.bias"] = torch.zeros(co, 2 * ci, 3, 3), torch.zeros(co)
            P[f"{b}.conv4.weight"], P[f"{b}.conv4.bias"] = torch.zeros(co, co, 3, 3), torch.zeros(co)
            P[f"{b}.conv1_3.weight"], P[f"{b}.conv1_3.bias"] = torch.zeros(ci, ci, 1, 1), torch.zeros(ci)
            P[f"{b}.conv3_4.weight"], P[f"{b}.conv3_4.bias"] = torch.zeros(co, 2 * ci, 1, 1), torch.zeros(co)
        P["final.weight"], P["final.bias"] = torch.zeros(out_channels, f[0], 1, 1), torch.zeros(out_channels)
        self._finish_init(seed)

    def init_weights(self, seed: int = None):
        g = self._gen(seed)            # lllinet.py:226-227: init_weights is a no-op -> torch defaults; r = 0.5, norm weight 1 / bias 0
        for k in list(self._params):
            if k.endswith(".weight") and self._params[k].dim() == 4:
                self._init_conv(k[:-7], g)
        self._on_params_change()

    def _block(self, b, srcs):
        c, ci = self._conv, self._io[b][0]
        cp = None if self.compute_dtype == "f32" or ci % 8 == 0 else (ci + 7) // 8 * 8     # conv0_0: 3 channels carried as 8 on the 16-bit path
        x1 = c(srcs, f"{b}.conv1", cout_pad=cp)
        if f"{b}.norm1.r" in self._params:
            x1 = ops.learnable_instance_norm(x1, self._vec(f"{b}.norm1.weight"), self._vec(f"{b}.norm1.bias"), self._vec(f"{b}.norm1.r"), 1e-5, "lrelu", 0.2)
        else:
            x1 = ops.channel_affine_act(x1, "lrelu", 0.2)
        x1 = ops.simam(x1, 1e-4)                                      # zero padding channels stay zero: x * sigmoid(.)
        x2 = c(x1, f"{b}.conv2", cout_pad=cp)
        s13 = c(srcs, f"{b}.conv1_3", pad=0, cout_pad=cp)
        v = None if cp is None else [ci, ci]
        x34 = c([x2, s13], f"{b}.conv3_4", pad=0, valid=v)
        x3 = c([x2, s13], f"{b}.conv3", act="lrelu", slope=0.2, valid=v)
        return c(x3, f"{b}.conv4", act="lrelu", slope=0.2, residual=x34)

    def forward(self, datapoint: dict, *args, **kwargs) -> dict:
        y = self._body(self._input_any(datapoint), clamp=(0.0, 1.0))
        return OrderedDict(enhanced=ops.to_nchw(y))

    def _body(self, x, clamp):
        """The UNet++ grid on the network input ``x`` (compute-dtype NHWC) -> ``final``'s output, fp32 NHWC, clamped if asked."""
        if x.shape[2] % 16 or x.shape[3] % 16:
            raise ValueError(f"LLLINet needs H and W divisible by 16 (got {x.shape[2]}x{x.shape[3]}); use infer(), which resizes to a "
                             f"multiple of 32 like the reference (vision/model.py:106).")
        B, up, pool = self._block, ops.upsample2x_bilinear, ops.maxpool2x2
        x0_0 = B("conv0_0", [x])
        x1_0 = B("conv1_0", [pool(x0_0)])
        x2_0 = B("conv2_0", [pool(x1_0)])
        x3_0 = B("conv3_0", [pool(x2_0)])
        x4_0 = B("conv4_0", [pool(x3_0)])
        x3_1 = B("conv3_1", [x3_0, up(x4_0)])
        u3_0 = up(x3_0)
        x2_1 = B("conv2_1", [x2_0, u3_0, up(x3_1)])
        u2_0 = up(x2_0)
        x1_1 = B("conv1_1", [x1_0, u2_0, up(x2_1)])
        u1_0 = up(x1_0)
        x0_1 = B("conv0_1", [x0_0, u1_0, up(x1_1)])
        x2_2 = B("conv2_2", [x2_0, x2_1, up(x3_1)])
        x1_2 = B("conv1_2", [x1_0, x1_1, up(x2_1), up(x2_2)])
        x0_2 = B("conv0_2", [x0_0, x0_1, up(x1_1), up(x1_2)])
        x1_3 = B("conv1_3", [x1_0, x1_1, x1_2, up(x2_2)])
        x0_3 = B("conv0_3", [x0_0, x0_1, x0_2, up(x1_2), up(x1_3)])
        x0_4 = B("conv0_4", [x0_0, x0_1, x0_2, x0_3, up(x1_3)])
        return self._conv(x0_4, "final", pad=0, clamp=clamp, out_f32=True)


@MODELS.register(name="lllinet_hvi", arch="lllinet")
class LLLINetHVI(LLLINet):
    """LLLINetHVI (lllinet.py:262-388): the LLLINet grid between ``core.RGBToHVI.rgb_to_hvi`` and ``hvi_to_rgb`` (core/image/color.py:313-434, HVI-CIDNet's colour
    space; ``ie_rgb_to_hvi_nhwc_f32`` / ``ie_hvi_to_rgb_nhwc_f32``), clamp to [0, 1] behind the inverse transform.  Extra checkpoint key ``trans.density_k`` (the
    transform's learnable exponent, 0.2); ``hvi_weight`` / ``loss_weights`` are training hyper-parameters read from the checkpoint's dict like the reference does."""

    def __init__(self, in_channels: int = 3, out_channels: int = 3, weights: Any = None, hvi_weight: float = 1.0, loss_weights=None, seed: int = None, *args, **kwargs):
        kwargs.pop("name", None)
        super().__init__(in_channels=in_channels, out_channels=out_channels, weights=weights, seed=seed, *args, **kwargs)
        self.name = self.fullname = "lllinet_hvi"
        if isinstance(self.weights, dict):
            hvi_weight = self.weights.get("hvi_weight", hvi_weight)
            loss_weights = self.weights.get("loss_weights", loss_weights)
        self.hvi_weight, self.loss_weights = hvi_weight, loss_weights or [0.35, 0.10, 0.25, 0.30]

    def _finish_init(self, seed):
        self._params["trans.density_k"] = torch.full([1], 0.2)          # before weights are loaded / drawn: it is a checkpoint key
        super()._finish_init(seed)

    def init_weights(self, seed: int = None):
        super().init_weights(seed)
        self._params["trans.density_k"] = torch.full([1], 0.2)

    def rgb_to_hvi(self, x):
        return ops.rgb_to_hvi(x, float(self._params["trans.density_k"].reshape(-1)[0]))

    def forward(self, datapoint: dict, *args, **kwargs) -> dict:
        k = float(self._params["trans.density_k"].reshape(-1)[0])
        hvi = ops.rgb_to_hvi(self._input(datapoint), k)                 # fp32 NHWC
        x = hvi if self.compute_dtype == "f32" else ops.cast(hvi, self._tdt, 8)
        y = self._body(x, clamp=None)
        return OrderedDict(enhanced=ops.to_nchw(ops.hvi_to_rgb(y, k, clamp01=True)))


# ----------------------------------------------------------------------------------------------------------------
@MODELS.register(name="mprnet", arch="mprnet")
class MPRNet(_GraphModel):
    """MPRNet (mprnet.py:349-548): three stages over a 4 / 2 / 1 patch hierarchy.  Stage 1 and 2 are encoder-decoders of channel
    attention blocks (CAB, :138-160: conv3x3 - PReLU - conv3x3, global-average-pool -> 1x1 -> ReLU -> 1x1 -> sigmoid channel gate,
    residual) with bilinear x0.5 / x2 + 1x1 resampling (:50-113) and cross-stage feature fusion (:205-238); stage 3 is the
    original-resolution ORSNet (:296-343); supervised attention modules (SAM, :163-182) hand the features on and emit the stage
    images.  The ONE ``nn.PReLU()`` of :421 is shared by all CABs: its slope is read from ``shallow_feat1.1.body.1.weight`` (a
    checkpoint repeats it under every ``...body.1.weight``).  Outputs ``stage1, stage2, enhanced`` (:543-547).  fp32 parity path and the
    16-bit matrix-core path (images as 8 channels, the channel gates' perceptron in fp32); H and W must be multiples of 8 (two patch
    splits and two x0.5 levels)."""

    arch = "mprnet"
    tasks = [Task.DEBLUR, Task.DENOISE, Task.DERAIN, Task.DESNOW]
    schemes = [Scheme.SUPERVISED]

    def __init__(self, in_channels: int = 3, num_channels: int = 96, scale_unetfeats: int = 48, scale_orsnetfeats: int = 32, num_cab: int = 8,
                 kernel_size: int = 3, reduction: int = 4, bias: bool = False, weights: Any = None, seed: int = None, *args, **kwargs):
        kwargs.pop("name", None)
        super().__init__(name="mprnet", in_channels=in_channels, weights=weights, *args, **kwargs)
        if isinstance(self.weights, dict):
            in_channels = self.weights.get("in_channels", in_channels)
            num_channels = self.weights.get("num_channels", num_channels)
            scale_unetfeats = self.weights.get("scale_unetfeats", scale_unetfeats)
            scale_orsnetfeats = self.weights.get("scale_orsnetfeats", scale_orsnetfeats)
            num_cab = self.weights.get("num_cab", num_cab)
            kernel_size = self.weights.get("kernel_size", kernel_size)
            reduction = self.weights.get("reduction", reduction)
            bias = self.weights.get("bias", bias)
        self.in_channels, self.num_channels, self.scale_unetfeats, self.scale_orsnetfeats = in_channels, num_channels, scale_unetfeats, scale_orsnetfeats
        self.num_cab, self.kernel_size, self.reduction, self.bias = num_cab, kernel_size, reduction, bias
        self.predicting = True
        P, k, c, su, so = self._params, kernel_size, num_channels, scale_unetfeats, scale_orsnetfeats

        def conv(name, ci, co, kk, b=bias):
            P[f"{name}.weight"] = torch.zeros(co, ci, kk, kk)
            if b:
                P[f"{name}.bias"] = torch.zeros(co)

        def cab(name, ch):
            conv(f"{name}.body.0", ch, ch, k)
            P[f"{name}.body.1.weight"] = torch.full((1,), 0.25)
            conv(f"{name}.body.2", ch, ch, k)
            conv(f"{name}.CA.conv_du.0", ch, ch // reduction, 1)
            conv(f"{name}.CA.conv_du.2", ch // reduction, ch, 1)

        for i in (1, 2, 3):
            conv(f"shallow_feat{i}.0", in_channels, c, k)
            cab(f"shallow_feat{i}.1", c)
        lv = [c, c + su, c + 2 * su]
        for st in (1, 2):
            e, d = f"stage{st}_encoder", f"stage{st}_decoder"
            for l in range(3):
                for i in range(2):
                    cab(f"{e}.encoder_level{l + 1}.{i}", lv[l])
                    cab(f"{d}.decoder_level{l + 1}.{i}", lv[l])
                if st == 2:
                    conv(f"{e}.csff_enc{l + 1}", lv[l], lv[l], 1)
                    conv(f"{e}.csff_dec{l + 1}", lv[l], lv[l], 1)
            conv(f"{e}.down12.down.1", lv[0], lv[1], 1, False)
            conv(f"{e}.down23.down.1", lv[1], lv[2], 1, False)
            cab(f"{d}.skip_attn1", lv[0])
            cab(f"{d}.skip_attn2", lv[1])
            conv(f"{d}.up21.up.1", lv[1], lv[0], 1, False)
            conv(f"{d}.up32.up.1", lv[2], lv[1], 1, False)
        o = "stage3_orsnet"
        for j in (1, 2, 3):
            for i in range(num_cab):
                cab(f"{o}.orb{j}.body.{i}", c + so)
            conv(f"{o}.orb{j}.body.{num_cab}", c + so, c + so, k, False)          # conv()'s default bias=False (:33-47, :284)
            conv(f"{o}.conv_enc{j}", c, c + so, 1)
            conv(f"{o}.conv_dec{j}", c, c + so, 1)
        for n in ("up_enc1", "up_dec1", "up_enc2.1", "up_dec2.1"):
            conv(f"{o}.{n}.up.1", lv[1], lv[0], 1, False)
        for n in ("up_enc2.0", "up_dec2.0"):
            conv(f"{o}.{n}.up.1", lv[2], lv[1], 1, False)
        for n in ("sam12", "sam23"):
            conv(f"{n}.conv1", c, c, 1)
            conv(f"{n}.conv2", c, 3, 1)
            conv(f"{n}.conv3", 3, c, 1)
        conv("concat12", 2 * c, c, k)
        conv("concat23", 2 * c, c + so, k)
        conv("tail", c + so, in_channels, k)
        self._finish_init(seed)

    def init_weights(self, seed: int = None):
        g = self._gen(seed)                # mprnet.py:444-445: init_weights is a no-op -> torch defaults (PReLU slope 0.25)
        for name in list(self._params):
            if name.endswith(".weight") and self._params[name].dim() == 4:
                self._init_conv(name[:-7], g)
        self._on_params_change()

    def _cv(self, t, name, **kw):
        w = self._params[f"{name}.weight"]
        if self.compute_dtype != "f32" and w.shape[0] % 8:               # 3-channel images travel as 8 channels on the 16-bit path
            kw.setdefault("cout_pad", (w.shape[0] + 7) // 8 * 8)
        return self._conv(t, name, pad=w.shape[2] // 2, **kw)

    def _cab(self, name, t):
        slope = float(self._params["shallow_feat1.1.body.1.weight"].reshape(-1)[0])
        y = self._cv(self._cv(t, f"{name}.body.0", act="lrelu", slope=slope), f"{name}.body.2")
        # the gate is a two-layer perceptron on the pooled vector: fp32 on every path (C/4 channels need not be a multiple of 8)
        a = ops.conv2d(ops.global_avg_pool(y), self._cw(f"{name}.CA.conv_du.0"), pad=0, act="relu")
        a = ops.conv2d(a, self._cw(f"{name}.CA.conv_du.2"), pad=0, act="sigmoid")
        return ops.channel_affine(y, a.reshape(a.shape[0], a.shape[1]), residual=t)          # y * gate + x (mprnet.py:156-161), one pass

    def _cabs(self, name, t, k=2):
        for i in range(k):
            t = self._cab(f"{name}.{i}", t)
        return t

    def _down(self, name, t):
        return self._cv(ops.resize(t, (t.shape[2] // 2, t.shape[3] // 2), "bilinear"), f"{name}.down.1")

    def _up(self, name, t, residual=None):
        return self._cv(ops.resize(t, (t.shape[2] * 2, t.shape[3] * 2), "bilinear"), f"{name}.up.1", residual=residual)

    def _encoder(self, n, t, enc=None, dec=None):
        outs = []
        for l in (1, 2, 3):
            if l > 1:
                t = self._down(f"{n}.down{l - 1}{l}", t)
            t = self._cabs(f"{n}.encoder_level{l}", t)
            if enc is not None:
                t = self._cv(dec[l - 1], f"{n}.csff_dec{l}", residual=self._cv(enc[l - 1], f"{n}.csff_enc{l}", residual=t))
            outs.append(t)
        return outs

    def _decoder(self, n, e):
        d3 = self._cabs(f"{n}.decoder_level3", e[2])
        d2 = self._cabs(f"{n}.decoder_level2", self._up(f"{n}.up32", d3, residual=self._cab(f"{n}.skip_attn2", e[1])))
        d1 = self._cabs(f"{n}.decoder_level1", self._up(f"{n}.up21", d2, residual=self._cab(f"{n}.skip_attn1", e[0])))
        return [d1, d2, d3]

    def _sam(self, n, t, img):
        im = self._cv(t, f"{n}.conv2", residual=img)
        gate = self._cv(im, f"{n}.conv3", act="sigmoid")
        return ops.add(ops.mul(self._cv(t, f"{n}.conv1"), gate), t), im

    def _orb(self, n, t):
        return self._cv(self._cabs(f"{n}.body", t, self.num_cab), f"{n}.body.{self.num_cab}", residual=t)

    def forward(self, datapoint: dict, *args, **kwargs) -> dict:
        x = self._input_any(datapoint)                # fp32 (N,3,H,W) or 16-bit (N,8,H,W) with zero channels
        h, w = x.shape[2], x.shape[3]
        if h % 8 or w % 8:
            raise ValueError(f"MPRNet needs H and W divisible by 8 (got {h}x{w}); use infer(), which resizes to a multiple of 32 like the "
                             f"reference (vision/model.py:106).")
        shallow = lambda k, t: self._cab(f"shallow_feat{k}.1", self._cv(t, f"shallow_feat{k}.0"))
        top, bot = ops.crop(x, 0, 0, h // 2, w), ops.crop(x, h // 2, 0, h - h // 2, w)
        quads = [ops.crop(x, 0, 0, h // 2, w // 2), ops.crop(x, 0, w // 2, h // 2, w - w // 2),
                 ops.crop(x, h // 2, 0, h - h // 2, w // 2), ops.crop(x, h // 2, w // 2, h - h // 2, w - w // 2)]
        f1 = [self._encoder("stage1_encoder", shallow(1, q)) for q in quads]
        f1_top = [ops.cat_spatial([a, b], 3) for a, b in zip(f1[0], f1[1])]
        f1_bot = [ops.cat_spatial([a, b], 3) for a, b in zip(f1[2], f1[3])]
        r1_top, r1_bot = self._decoder("stage1_decoder", f1_top), self._decoder("stage1_decoder", f1_bot)
        s_top, img_top = self._sam("sam12", r1_top[0], top)
        s_bot, img_bot = self._sam("sam12", r1_bot[0], bot)
        stage1 = ops.cat_spatial([img_top, img_bot], 2)
        c_top = self._cv([shallow(2, top), s_top], "concat12")
        c_bot = self._cv([shallow(2, bot), s_bot], "concat12")
        f2_top = self._encoder("stage2_encoder", c_top, f1_top, r1_top)
        f2_bot = self._encoder("stage2_encoder", c_bot, f1_bot, r1_bot)
        f2 = [ops.cat_spatial([a, b], 2) for a, b in zip(f2_top, f2_bot)]
        r2 = self._decoder("stage2_decoder", f2)
        s3, stage2 = self._sam("sam23", r2[0], x)
        t = self._cv([shallow(3, x), s3], "concat23")
        o = "stage3_orsnet"
        t = self._orb(f"{o}.orb1", t)
        t = self._cv(r2[0], f"{o}.conv_dec1", residual=self._cv(f2[0], f"{o}.conv_enc1", residual=t))
        t = self._orb(f"{o}.orb2", t)
        t = self._cv(self._up(f"{o}.up_dec1", r2[1]), f"{o}.conv_dec2", residual=self._cv(self._up(f"{o}.up_enc1", f2[1]), f"{o}.conv_enc2", residual=t))
        t = self._orb(f"{o}.orb3", t)
        e3 = self._up(f"{o}.up_enc2.1", self._up(f"{o}.up_enc2.0", f2[2]))
        d3 = self._up(f"{o}.up_dec2.1", self._up(f"{o}.up_dec2.0", r2[2]))
        t = self._cv(d3, f"{o}.conv_dec3", residual=self._cv(e3, f"{o}.conv_enc3", residual=t))
        y = self._cv(t, "tail", residual=x)
        if self.compute_dtype != "f32":                # drop the padding channels, back to fp32
            stage1, stage2, y = (ops.cast(v, torch.float32, self.in_channels) for v in (stage1, stage2, y))
        return OrderedDict(stage1=ops.to_nchw(stage1), stage2=ops.to_nchw(stage2), enhanced=ops.to_nchw(y))


# ----------------------------------------------------------------------------------------------------------------
@MODELS.register(name="unet", arch="unet")
class UNet(_GraphModel):
    """unet.py:115-197 with its intended semantics (line 51's ``self.con1`` is a typo for ``conv1``): (Conv3x3 no-bias ->
    BatchNorm2d -> ReLU) x 2 blocks, MaxPool2d(2), bilinear x2 (align_corners=True) + zero pad + cat([skip, up]),
    1x1 ``outc``.  BatchNorm runs in eval mode (running statistics) folded into the conv weights.

    Engine extras (BASELINE config 3/5): ``base_channels`` (64 -> U-Net-64, 32 -> U-Net-32), ``upsampler``
    ``"bilinear"`` (reference default, unet.py:82-83), ``"convtranspose"`` (the reference's ``bilinear=False`` branch, unet.py:85-86:
    ``upK.up`` = ConvTranspose2d(in, in // 2, 2, 2), un-halved widths) or ``"pixelshuffle"`` (conv3x3(C -> 4*C/2... ) +
    PixelShuffle(2), edsr.py:88-123), ``enhance=True`` returns ``enhanced = clamp(outc, 0, 1)`` (llunetpp.py:241-242) instead of
    ``semantic``, ``conv_bias=True`` gives the 3x3 convs a bias like the original modules the reference tree also ships
    (mon_extra/vision/enhance/llie/mtfe/unet_parts.py:15-21); checkpoints in that file's key spelling (``*.double_conv.{0,1,3,4}``,
    ``downK.maxpool_conv.1``) load directly."""

    arch = "unet"
    tasks = [Task.SEGMENT, Task.LLIE]
    schemes = [Scheme.SUPERVISED]

    def __init__(self, in_channels: int = 3, num_classes: int = 34, weights: Any = None, base_channels: int = 64,
                 upsampler: str = "bilinear", enhance: bool = False, conv_bias: bool = False, seed: int = None, *args, **kwargs):
        kwargs.pop("name", None)
        super().__init__(name="unet", in_channels=in_channels, num_classes=num_classes, weights=weights, *args, **kwargs)
        if isinstance(self.weights, dict):
            in_channels = self.weights.get("in_channels", in_channels)
            num_classes = self.weights.get("num_classes", num_classes)
        if upsampler not in ("bilinear", "pixelshuffle", "convtranspose"):
            raise ValueError(f"`upsampler` must be 'bilinear', 'convtranspose' or 'pixelshuffle', but got {upsampler}.")
        self.in_channels, self.out_channels = in_channels, num_classes
        self.base_channels, self.upsampler, self.enhance, self.conv_bias = base_channels, upsampler, enhance, bool(conv_bias)
        self.predicting = True
        f = [base_channels * m for m in (1, 2, 4, 8, 16)]
        self._f = f
        factor = 1 if upsampler == "convtranspose" else 2      # unet.py:153: factor = 2 if bilinear else 1

        def dc(name, ci, co, mid=None):
            mid = mid or co
            P = self._params
            P[f"{name}.conv1.weight"] = torch.zeros(mid, ci, 3, 3)
            P[f"{name}.conv2.weight"] = torch.zeros(co, mid, 3, 3)
            if self.conv_bias:
                P[f"{name}.conv1.bias"], P[f"{name}.conv2.bias"] = torch.zeros(mid), torch.zeros(co)
            for nm, c in (("norm1", mid), ("norm2", co)):
                P[f"{name}.{nm}.weight"], P[f"{name}.{nm}.bias"] = torch.ones(c), torch.zeros(c)
                P[f"{name}.{nm}.running_mean"], P[f"{name}.{nm}.running_var"] = torch.zeros(c), torch.ones(c)

        dc("inc", in_channels, f[0])
        dc("down1.conv", f[0], f[1]); dc("down2.conv", f[1], f[2]); dc("down3.conv", f[2], f[3]); dc("down4.conv", f[3], f[4] // factor)
        for k, (ci, co) in zip((1, 2, 3, 4), ((f[4], f[3] // factor), (f[3], f[2] // factor), (f[2], f[1] // factor), (f[1], f[0]))):
            if upsampler == "convtranspose":      # unet.py:85-86: ConvTranspose2d(in, in // 2, 2, 2) + DoubleConv(in, out)
                self._params[f"up{k}.up.weight"], self._params[f"up{k}.up.bias"] = torch.zeros(ci, ci // 2, 2, 2), torch.zeros(ci // 2)
                dc(f"up{k}.conv", ci, co)
            else:                                 # unet.py:82-83: DoubleConv(in, out, mid = in // 2)
                dc(f"up{k}.conv", ci, co, ci // 2)
        if upsampler == "pixelshuffle":       # Upsampler(scale 2): conv3x3(C -> 4C) + PixelShuffle(2), per decoder level
            for k, c in zip((1, 2, 3, 4), (f[4] // 2, f[3] // 2, f[2] // 2, f[1] // 2)):
                self._params[f"up{k}.up.0.weight"] = torch.zeros(4 * c, c, 3, 3)
                self._params[f"up{k}.up.0.bias"] = torch.zeros(4 * c)
        self._params["outc.conv.weight"], self._params["outc.conv.bias"] = torch.zeros(num_classes, f[0], 1, 1), torch.zeros(num_classes)
        self._finish_init(seed)

    _SEQ_KEYS = {"0": "conv1", "1": "norm1", "3": "conv2", "4": "norm2"}

    def map_key(self, k: str) -> str:
        """The original U-Net modules' spelling (mon_extra/vision/enhance/llie/mtfe/unet_parts.py:8-40: ``double_conv`` Sequential
        with conv / BN at 0,1,3,4 and ``downK.maxpool_conv.1``) -> mon's (unet.py:42-47, 66)."""
        parts = k.replace(".maxpool_conv.1.", ".conv.").split(".")
        if "double_conv" in parts:
            i = parts.index("double_conv")
            if i + 1 < len(parts) and parts[i + 1] in self._SEQ_KEYS:
                parts[i:i + 2] = [self._SEQ_KEYS[parts[i + 1]]]
        return ".".join(parts)

    def conv_transpose_keys(self):
        return [k for k in self._params if k.endswith(".up.weight") and self.upsampler == "convtranspose"]

    def load_state_dict(self, state_dict: dict, strict: bool = True):
        sd = {k: v for k, v in state_dict.items() if not k.endswith("num_batches_tracked")}
        super().load_state_dict(sd, strict)

    def init_weights(self, seed: int = None):
        g = self._gen(seed)
        for k in list(self._params):
            if k.endswith(".weight") and self._params[k].dim() == 4:
                self._init_conv(k[:-7], g)        # (a ConvTranspose2d weight is (Cin, Cout, 2, 2): torch's fan_in for it is Cout * 4 = shape[1] * 4 too)
            elif k.endswith("running_mean"):      # seeded non-trivial BN statistics so that eval-mode BN is exercised
                self._params[k] = torch.randn(self._params[k].shape, generator=g) * 0.1
            elif k.endswith("running_var"):
                self._params[k] = torch.rand(self._params[k].shape, generator=g) + 0.5
            elif ".norm" in k and k.endswith(".weight"):
                self._params[k] = 1.0 + torch.randn(self._params[k].shape, generator=g) * 0.02
        self._on_params_change()

    def _dc(self, name, srcs, pool=False):
        """DoubleConv; ``pool=True`` also returns the MaxPool2d(2) the next ``Down`` starts with (unet.py:70-72)."""
        if pool and self.compute_dtype != "f32" and isinstance(srcs, (list, tuple)) and len(srcs) == 1 and int(srcs[0].shape[1]) == 32:
            mwa = self._mw_plain((32,), f"{name}.conv1", f"{name}.norm1")          # 32 -> 64 -> 64 + pool: one launch, the intermediate in LDS only
            mwb = self._mw_plain((mwa.cout,), f"{name}.conv2", f"{name}.norm2")
            if ops.mconv2d_pair_pool_applicable(srcs[0], mwa, mwb):
                return ops.mconv2d_pair_pool(srcs[0], mwa, mwb)
        x = self._conv(srcs, f"{name}.conv1", f"{name}.norm1", act="relu")
        return self._conv(x, f"{name}.conv2", f"{name}.norm2", act="relu", pool=pool)

    def _dc_head(self, name, srcs, clamp):
        """The last DoubleConv + ``outc`` (1x1) + clamp -> NCHW fp32; conv2 and outc are one launch where the strip kernel covers conv2."""
        x = self._conv(srcs, f"{name}.conv1", f"{name}.norm1", act="relu")
        y = self._conv_tail(x, f"{name}.conv2", f"{name}.norm2", "outc.conv", clamp=clamp)
        if y is not None:
            return y
        x = self._conv(x, f"{name}.conv2", f"{name}.norm2", act="relu")
        return ops.to_nchw(self._conv(x, "outc.conv", pad=0, clamp=clamp, out_f32=True))

    def _up_head_fused(self, k, x1, x2, clamp):
        """The last decoder level + ``outc`` as one launch (ie_unet_up_head_nhwc): DoubleConv over cat([skip, up]) and the 1x1 head; with the bilinear
        upsampler and an exactly doubled size the x2 up-sampling happens inside too (``x1`` goes in at half resolution).  None where it does not apply."""
        if self.compute_dtype == "f32" or self._params["outc.conv.weight"].shape[0] > 16:
            return None
        name = f"up{k}.conv"
        c = int(x2.shape[1])
        if int(x1.shape[1]) != c or tuple(self._params[f"{name}.conv1.weight"].shape[:2]) != (c, 2 * c):
            return None
        mwa = self._mw_plain((c, c), f"{name}.conv1", f"{name}.norm1")
        mwb = self._mw_plain((c,), f"{name}.conv2", f"{name}.norm2")
        tkey = ("mtail", "outc.conv")
        if tkey not in self._dev_cache:
            self._dev_cache[tkey] = ops.MTailWeights(self._params["outc.conv.weight"], self._params.get("outc.conv.bias"), self._tdt, self.device)
        tw = self._dev_cache[tkey]
        if not ops.unet_up_head_applicable(x2, x1, mwa, mwb, tw):
            return None
        return ops.unet_up_head(x2, x1, mwa, mwb, tw, clamp=clamp)

    def _up(self, k, x1, x2, head=False):
        clamp = (0.0, 1.0) if self.enhance else None
        if head and self.upsampler == "bilinear":         # half-resolution x1 straight into the fused level (where the kernel covers the in-kernel up-sampling)
            y = self._up_head_fused(k, x1, x2, clamp)
            if y is not None:
                return y
        if self.upsampler == "bilinear":
            dy, dx = x2.shape[2] - 2 * x1.shape[2], x2.shape[3] - 2 * x1.shape[3]
            if self.compute_dtype != "f32" and (dy > 0 or dx > 0) and dy >= 0 and dx >= 0:       # odd skip size: up-sample straight into the padded tensor
                x1 = ops.resize_pad(x1, (2 * x1.shape[2], 2 * x1.shape[3]), (dy // 2, dy - dy // 2, dx // 2, dx - dx // 2))
            else:
                x1 = ops.upsample2x_bilinear(x1)
        elif self.upsampler == "convtranspose":
            x1 = self._conv_transpose2x2(x1, f"up{k}.up")
        else:
            x1 = self._conv(x1, f"up{k}.up.0", shuffle=2)           # conv3x3 + PixelShuffle(2), fused into the conv's epilogue on the 16-bit path
        dy, dx = x2.shape[2] - x1.shape[2], x2.shape[3] - x1.shape[3]
        x1 = ops.pad(x1, dy // 2, dy - dy // 2, dx // 2, dx - dx // 2)
        if head:
            y = self._up_head_fused(k, x1, x2, clamp)
            if y is not None:
                return y
            return self._dc_head(f"up{k}.conv", [x2, x1], clamp=clamp)
        return self._dc(f"up{k}.conv", [x2, x1])

    def _first_level_nchw(self, datapoint):
        """``inc`` + the first max-pool straight from the fp32 NCHW image (ie_unet_first_pool_nchw, 32 channels), or None."""
        if self.compute_dtype == "f32" or self._f[0] != 32 or self.in_channels != 3:
            return None
        x = datapoint.get("image")
        if not isinstance(x, torch.Tensor):
            return None
        x = x[None] if x.dim() == 3 else x
        if x.dim() != 4 or x.shape[1] != 3:
            return None
        self._require_cuda()
        key = ("unetfirst", "inc.conv1")
        if self._dev_cache is None:
            self._dev_cache = {}
        if key not in self._dev_cache:
            self._dev_cache[key] = ops.UNetFirstWeights(*self._wb("inc.conv1", "inc.norm1"), self._tdt, self.device)
        w1, mw2 = self._dev_cache[key], self._mw_plain((32,), "inc.conv2", "inc.norm2")
        x = x.to(self.device, torch.float32)
        if not ops.unet_first_pool_applicable(x, w1, mw2):
            return None
        return ops.unet_first_pool(x, w1, mw2)

    def forward(self, datapoint: dict, *args, **kwargs) -> dict:
        first = self._first_level_nchw(datapoint)
        if first is not None:
            self.assert_datapoint(datapoint)
            x1, p1 = first
            return self._after_first_level(x1, p1)
        x = self._input_any(datapoint)
        x1 = None
        if self.compute_dtype != "f32" and x.shape[1] == 8:       # the first DoubleConv + its pool as one launch (conv1's output stays in the kernel)
            mw1 = self._mw_plain((8,), "inc.conv1", "inc.norm1")
            mw2 = self._mw_plain((self._f[0],), "inc.conv2", "inc.norm2")
            if ops.mconv2d_first_pool_applicable(x, mw1, mw2):
                x1, p1 = ops.mconv2d_first_pool(x, mw1, mw2)
        if x1 is None:
            x1, p1 = self._dc("inc", [x], pool=True)
        return self._after_first_level(x1, p1)

    def _after_first_level(self, x1, p1):
        x2, p2 = self._dc("down1.conv", [p1], pool=True)
        x3, p3 = self._dc("down2.conv", [p2], pool=True)
        x4, p4 = self._dc("down3.conv", [p3], pool=True)
        x5 = self._dc("down4.conv", [p4])
        y = self._up(1, x5, x4)
        y = self._up(2, y, x3)
        y = self._up(3, y, x2)
        y = self._up(4, y, x1, head=True)                 # ... + outc + clamp, NCHW fp32
        out = OrderedDict()
        out["enhanced" if self.enhance else "semantic"] = y
        return out

    supports_mixed_training = True

    def forward_train(self, x, store, tape):
        """Training-mode graph (BatchNorm on batch statistics, running stats updated; activations kept on ``tape``):
        what ``model.train(); model(x)`` runs in the reference (unet.py:35-197).  Returns the NHWC prediction."""
        from . import train as T

        mixed = store.compute_dtype != "f32"
        tdt = ops.TORCH_DT[store.compute_dtype]

        def dc(name, srcs):
            y = None
            for i in (1, 2):
                cn, nn_ = f"{name}.conv{i}", f"{name}.norm{i}"
                z = T.t_conv(tape, srcs, store.conv(cn), *store.conv_grads(cn))
                y = T.t_norm(tape, z, store.view(f"{nn_}.weight"), store.view(f"{nn_}.bias"), store.view(f"{nn_}.weight", store.grads),
                             store.view(f"{nn_}.bias", store.grads), per_sample=False, running_mean=store.buffers[f"{nn_}.running_mean"],
                             running_var=store.buffers[f"{nn_}.running_var"], momentum=0.1, eps=1e-5, act="relu")
                srcs = [y]
            return y

        def up(k, a, skip):
            if self.upsampler == "bilinear":
                a = T.t_resize(tape, a, (a.shape[2] * 2, a.shape[3] * 2), "bilinear_align")
            elif self.upsampler == "convtranspose":      # on the 16-bit graph: the 1x1 matrix-core conv to 4*Cout channels + pixel shuffle (IE_CONVT_FP32=1: the fp32 detour)
                if mixed and os.environ.get("IE_CONVT_FP32") == "1":
                    a = T.t_cast(tape, T.t_conv_transpose2x2(tape, T.t_cast(tape, a, torch.float32), store, f"up{k}.up"), tdt)
                else:
                    a = T.t_conv_transpose2x2(tape, a, store, f"up{k}.up")
            else:
                a = T.t_pixel_shuffle(tape, T.t_conv(tape, a, store.conv(f"up{k}.up.0"), *store.conv_grads(f"up{k}.up.0")), 2)
            dy, dx = skip.shape[2] - a.shape[2], skip.shape[3] - a.shape[3]
            a = T.t_pad(tape, a, dy // 2, dy - dy // 2, dx // 2, dx - dx // 2)
            return dc(f"up{k}.conv", [skip, a])

        if mixed:
            x = T.t_cast(tape, x, tdt, channels=16)    # RGB carried as 16 channels: the first layer runs on the matrix cores too
        x1 = dc("inc", [x])
        x2 = dc("down1.conv", [T.t_maxpool(tape, x1)])
        x3 = dc("down2.conv", [T.t_maxpool(tape, x2)])
        x4 = dc("down3.conv", [T.t_maxpool(tape, x3)])
        x5 = dc("down4.conv", [T.t_maxpool(tape, x4)])
        # gradient bucket boundary ("overlapped with wgrad of earlier layers"): everything from here on is the decoder + head,
        # whose parameters sit behind the encoder's in the flat buffer and whose gradients are complete first in the backward pass
        dec = [o for k, (o, _, _, _) in store.index.items() if k.startswith(("up", "outc"))]
        enc = [o for k, (o, _, _, _) in store.index.items() if not k.startswith(("up", "outc"))]
        if dec and enc and min(dec) > max(enc):
            tape.mark_grad_bucket(min(dec))
        y = up(1, x5, x4); y = up(2, y, x3); y = up(3, y, x2); y = up(4, y, x1)
        if mixed:                                      # head computed as 16 output channels; the clamp and the loss run in fp32
            y = T.t_conv(tape, y, store.conv("outc.conv"), *store.conv_grads("outc.conv"), pad=0, cout_pad=16)
            y = T.t_cast(tape, y, torch.float32, channels=self.out_channels)
        else:
            y = T.t_conv(tape, y, store.conv("outc.conv"), *store.conv_grads("outc.conv"), pad=0)
        return T.t_clamp(tape, y, 0.0, 1.0) if self.enhance else y

    def assert_outputs(self, outputs: dict):
        if not self.enhance:
            return
        super().assert_outputs(outputs)


# ----------------------------------------------------------------------------------------------------------------
@MODELS.register(name="zero_ig_enhance", arch="zero_ig")
class ZeroIGEnhance(_GraphModel):
    """``Enhance`` of Zero-IG (zero_ig.py:307-348): Conv3x3+ReLU, ``layers`` x [fea + ReLU(BN(Conv3x3(fea)))] with ONE
    shared block (zero_ig.py:324-326), Conv3x3+Sigmoid, clamp(1e-4, 1).  Keys ``in_conv.0.*``, ``conv.0.*``,
    ``conv.1.*`` (BN), ``out_conv.0.*`` (``blocks.i.*`` aliases of the shared block are accepted)."""

    arch = "zero_ig"
    tasks = [Task.LLIE]
    schemes = [Scheme.ZERO_SHOT]

    def __init__(self, layers: int = 3, channels: int = 64, weights: Any = None, seed: int = None, *args, **kwargs):
        kwargs.pop("name", None)
        super().__init__(name="zero_ig_enhance", in_channels=3, weights=weights, *args, **kwargs)
        self.layers, self.channels = layers, channels
        P = self._params
        P["in_conv.0.weight"], P["in_conv.0.bias"] = torch.zeros(channels, 3, 3, 3), torch.zeros(channels)
        P["conv.0.weight"], P["conv.0.bias"] = torch.zeros(channels, channels, 3, 3), torch.zeros(channels)
        P["conv.1.weight"], P["conv.1.bias"] = torch.ones(channels), torch.zeros(channels)
        P["conv.1.running_mean"], P["conv.1.running_var"] = torch.zeros(channels), torch.ones(channels)
        P["out_conv.0.weight"], P["out_conv.0.bias"] = torch.zeros(3, channels, 3, 3), torch.zeros(3)
        self._finish_init(seed)

    def load_state_dict(self, state_dict: dict, strict: bool = True):
        sd = {k: v for k, v in state_dict.items() if not k.endswith("num_batches_tracked") and not k.startswith("blocks.")}
        super().load_state_dict(sd, strict)

    def init_weights(self, seed: int = None):
        g = self._gen(seed)            # zero_ig.py:336-342: conv N(0, 0.02), bias 0, BN weight N(1, 0.02)
        for n in ("in_conv.0", "conv.0", "out_conv.0"):
            self._init_conv(n, g, std=0.02, zero_bias=True)
        self._params["conv.1.weight"] = 1.0 + torch.randn(self.channels, generator=g) * 0.02
        self._params["conv.1.running_mean"] = torch.randn(self.channels, generator=g) * 0.05
        self._params["conv.1.running_var"] = torch.rand(self.channels, generator=g) + 0.5
        self._on_params_change()

    def forward(self, datapoint: dict, *args, **kwargs) -> dict:
        x = self._input_any(datapoint)
        fea = self._conv(x, "in_conv.0", act="relu")
        for _ in range(self.layers):
            fea = self._conv(fea, "conv.0", "conv.1", act="relu", residual=fea)
        y = self._conv(fea, "out_conv.0", act="sigmoid", clamp=(0.0001, 1.0), out_f32=True)
        return OrderedDict(enhanced=ops.to_nchw(y))


# ----------------------------------------------------------------------------------------------------------------
@MODELS.register(name="zero_ig_re", arch="zero_ig")
class ZeroIG(_GraphModel):
    """Zero-IG (zero_ig.py:355-452), predicting branch of ``forward`` (:434-452 == ``Finetunemodel.forward`` of the original
    repository, mon_extra/.../zero_ig/model.py:188-199):

        image += 1e-4;  L2 = clamp(image - Denoise1(image), 1e-4, 1);  s2 = Enhance(L2);  H2 = clamp(image / s2, 1e-4, 1)
        H3 = clamp(cat[H2, s2] - Denoise2(cat[H2, s2]), 1e-4, 1)[:, :3]          -> {"denoise": H3, "enhanced": H2}

    ``Enhance`` (:307-348) is the graph of ``zero_ig_enhance``; ``Denoise1/2`` (:257-304) are Conv3x3 + LeakyReLU(0.2) twice and
    a Conv1x1.  ``x - conv(x)`` runs as ONE convolution with negated weights and the fused residual add + clamp.
    Checkpoint keys ``enhance.*``, ``denoise1.*``, ``denoise2.*`` (the original's ``denoise_1`` / ``denoise_2`` are accepted;
    the ``enhance.blocks.i.*`` aliases of the shared block are dropped)."""

    arch = "zero_ig"
    tasks = [Task.LLIE]
    schemes = [Scheme.UNSUPERVISED, Scheme.ZERO_REFERENCE]

    def __init__(self, in_channels: int = 3, num_channels: int = 64, embed_channels: int = 48, weights: Any = None, seed: int = None,
                 *args, **kwargs):
        kwargs.pop("name", None)
        super().__init__(name="zero_ig_re", in_channels=in_channels, weights=weights, *args, **kwargs)
        if isinstance(self.weights, dict):
            in_channels = self.weights.get("in_channels", in_channels)
            num_channels = self.weights.get("num_channels", num_channels)
            embed_channels = self.weights.get("embed_channels", embed_channels)
        self.in_channels, self.num_channels, self.embed_channels = in_channels, num_channels, embed_channels
        self.layers = in_channels                      # zero_ig.py:388: Enhance(layers=self.in_channels, ...)
        self.predicting = True
        c, e, P = num_channels, embed_channels, self._params
        P["enhance.in_conv.0.weight"], P["enhance.in_conv.0.bias"] = torch.zeros(c, 3, 3, 3), torch.zeros(c)
        P["enhance.conv.0.weight"], P["enhance.conv.0.bias"] = torch.zeros(c, c, 3, 3), torch.zeros(c)
        P["enhance.conv.1.weight"], P["enhance.conv.1.bias"] = torch.ones(c), torch.zeros(c)
        P["enhance.conv.1.running_mean"], P["enhance.conv.1.running_var"] = torch.zeros(c), torch.ones(c)
        P["enhance.out_conv.0.weight"], P["enhance.out_conv.0.bias"] = torch.zeros(3, c, 3, 3), torch.zeros(3)
        for n, ci in (("denoise1", 3), ("denoise2", 6)):
            P[f"{n}.conv1.weight"], P[f"{n}.conv1.bias"] = torch.zeros(e, ci, 3, 3), torch.zeros(e)
            P[f"{n}.conv2.weight"], P[f"{n}.conv2.bias"] = torch.zeros(e, e, 3, 3), torch.zeros(e)
            P[f"{n}.conv3.weight"], P[f"{n}.conv3.bias"] = torch.zeros(ci, e, 1, 1), torch.zeros(ci)
        self._finish_init(seed)

    def map_key(self, k: str) -> str:
        return k.replace("denoise_1.", "denoise1.").replace("denoise_2.", "denoise2.")

    def load_state_dict(self, state_dict: dict, strict: bool = True):
        sd = {k: v for k, v in state_dict.items() if not k.endswith("num_batches_tracked") and ".blocks." not in k
              and not k.startswith("texture_difference.")}
        super().load_state_dict(sd, strict)

    def init_weights(self, seed: int = None):
        g = self._gen(seed)            # zero_ig.py:404-409: conv N(0, 0.02), bias 0, BN weight N(1, 0.02)
        for k in list(self._params):
            if k.endswith(".weight") and self._params[k].dim() == 4:
                self._init_conv(k[:-7], g, std=0.02, zero_bias=True)
        c = self.num_channels
        self._params["enhance.conv.1.weight"] = 1.0 + torch.randn(c, generator=g) * 0.02
        self._params["enhance.conv.1.running_mean"] = torch.randn(c, generator=g) * 0.05
        self._params["enhance.conv.1.running_var"] = torch.rand(c, generator=g) + 0.5
        self._on_params_change()

    def forward(self, datapoint: dict, *args, **kwargs) -> dict:
        eps, c = 1e-4, self._conv
        f32 = self.compute_dtype == "f32"
        image = ops.affine(self._input(datapoint), 1.0, eps)                              # image + eps (fp32)
        img = image if f32 else ops.cast(image, self._tdt, 8)
        cp = None if f32 else 8

        def denoise_residual(n, srcs, res, cout, valid=None):
            # res - Denoise(srcs)[:, :cout], clamped to [eps, 1]
            y = c(srcs, f"{n}.conv1", act="lrelu", slope=0.2, valid=valid)
            y = c(y, f"{n}.conv2", act="lrelu", slope=0.2)
            return c(y, f"{n}.conv3", pad=0, negate=True, cout=cout, residual=res, clamp=(eps, 1.0), cout_pad=cp)

        l2 = denoise_residual("denoise1", img, img, 3)
        fea = c(l2, "enhance.in_conv.0", act="relu")
        for _ in range(self.layers):
            fea = c(fea, "enhance.conv.0", "enhance.conv.1", act="relu", residual=fea)
        s2 = c(fea, "enhance.out_conv.0", act="sigmoid", clamp=(0.0001, 1.0), out_f32=True)   # 3 channels, fp32 on every path
        h2 = ops.div_clamp(image, s2, eps, 1.0)
        if f32:
            h3 = denoise_residual("denoise2", [h2, s2], h2, 3)
        else:
            h2p, s2p = ops.cast(h2, self._tdt, 8), ops.cast(s2, self._tdt, 8)
            h3 = ops.cast(denoise_residual("denoise2", [h2p, s2p], h2p, 3, valid=[3, 3]), torch.float32, 3)
        return OrderedDict(denoise=ops.to_nchw(h3), enhanced=ops.to_nchw(h2))


# ----------------------------------------------------------------------------------------------------------------
@MODELS.register(name="rrdnet_re", arch="rrdnet")
class RRDNet_RE(_GraphModel):
    """RRDNet (rrdnet.py:183-289): three 5-layer Conv3x3 stacks (3-16-32-64-32-out, ReLU between) for illumination (1 ch,
    sigmoid), reflectance (3 ch, sigmoid) and noise (3 ch, tanh); ``enhanced = clamp(illumination^gamma * ((image - noise) /
    illumination), 0, 1)`` (:271-279).  Keys ``{illumination,reflectance,noise}_net.{0,2,4,6,8}.*``.  The reference's ``infer``
    fits the three stacks to ONE image for 1000 iterations (:291-335); this class provides ``forward``."""

    arch = "rrdnet"
    tasks = [Task.LLIE]
    schemes = [Scheme.UNSUPERVISED, Scheme.ZERO_REFERENCE, Scheme.INSTANCE]
    _NETS = (("illumination_net", 1, "sigmoid"), ("reflectance_net", 3, "sigmoid"), ("noise_net", 3, "tanh"))

    def __init__(self, gamma: float = 0.4, illu_factor: float = 1, reflect_factor: float = 1, noise_factor: float = 5000,
                 weights: Any = None, seed: int = None, *args, **kwargs):
        kwargs.pop("name", None)
        super().__init__(name="rrdnet_re", in_channels=3, weights=weights, *args, **kwargs)
        self.gamma, self.illu_factor, self.reflect_factor, self.noise_factor = gamma, illu_factor, reflect_factor, noise_factor
        self.predicting = True
        for net, co, _ in self._NETS:
            for i, (a, b) in zip((0, 2, 4, 6, 8), ((3, 16), (16, 32), (32, 64), (64, 32), (32, co))):
                self._params[f"{net}.{i}.weight"], self._params[f"{net}.{i}.bias"] = torch.zeros(b, a, 3, 3), torch.zeros(b)
        self._finish_init(seed)

    def init_weights(self, seed: int = None):
        g = self._gen(seed)            # rrdnet.py:255-256: init_weights is a no-op -> torch's Conv2d default
        for k in list(self._params):
            if k.endswith(".weight"):
                self._init_conv(k[:-7], g)
        self._on_params_change()

    def forward(self, datapoint: dict, *args, **kwargs) -> dict:
        image = self._input(datapoint)
        x = image if self.compute_dtype == "f32" else ops.cast(image, self._tdt, 8)
        outs = {}
        for net, co, act in self._NETS:
            y = x
            for i in (0, 2, 4, 6):
                y = self._conv(y, f"{net}.{i}", act="relu")
            outs[net] = self._conv(y, f"{net}.8", act=act, out_f32=True)
        illumination, reflectance, noise = outs["illumination_net"], outs["reflectance_net"], outs["noise_net"]
        adjust = ops.pow_scalar(illumination, self.gamma)
        enhanced = ops.clamp(ops.mul(ops.div_clamp(ops.sub(image, noise), illumination), adjust), 0.0, 1.0)
        out = OrderedDict(image=ops.to_nchw(image), illumination=ops.to_nchw(illumination), reflectance=ops.to_nchw(reflectance),
                          noise=ops.to_nchw(noise))
        out["enhanced"] = ops.to_nchw(enhanced)
        return out


# ----------------------------------------------------------------------------------------------------------------
@MODELS.register(name="zsn2n", arch="zsn2n")
class ZSN2N(_GraphModel):
    """ZS-N2N (zsn2n.py:40-124): Conv3x3 + LeakyReLU(0.2), Conv3x3 + LeakyReLU(0.2), Conv1x1; ``enhanced`` is clamped to [0, 1]
    when ``predicting`` (:116-124).  Keys ``conv1/conv2/conv3``.  (Its ``infer`` trains on the single input image for 3000
    iterations, :126-190; this class provides ``forward``.)"""

    arch = "zsn2n"
    tasks = [Task.DENOISE]
    schemes = [Scheme.UNSUPERVISED, Scheme.ZERO_SHOT, Scheme.INSTANCE]

    def __init__(self, in_channels: int = 3, num_channels: int = 48, weights: Any = None, seed: int = None, predicting: bool = True,
                 *args, **kwargs):
        kwargs.pop("name", None)
        super().__init__(name="zsn2n", in_channels=in_channels, weights=weights, *args, **kwargs)
        if isinstance(self.weights, dict):
            in_channels = self.weights.get("in_channels", in_channels)
            num_channels = self.weights.get("num_channels", num_channels)
        self.in_channels, self.num_channels, self.predicting = in_channels, num_channels, predicting
        c, P = num_channels, self._params
        P["conv1.weight"], P["conv1.bias"] = torch.zeros(c, in_channels, 3, 3), torch.zeros(c)
        P["conv2.weight"], P["conv2.bias"] = torch.zeros(c, c, 3, 3), torch.zeros(c)
        P["conv3.weight"], P["conv3.bias"] = torch.zeros(self.out_channels, c, 1, 1), torch.zeros(self.out_channels)
        self._finish_init(seed)

    def init_weights(self, seed: int = None):
        g = self._gen(seed)            # zsn2n.py:88-89: no-op -> torch's Conv2d default
        for n in ("conv1", "conv2", "conv3"):
            self._init_conv(n, g)
        self._on_params_change()

    def forward(self, datapoint: dict, *args, **kwargs) -> dict:
        x = self._input_any(datapoint)
        y = self._conv(x, "conv1", act="lrelu", slope=0.2)
        y = self._conv(y, "conv2", act="lrelu", slope=0.2)
        y = self._conv(y, "conv3", pad=0, clamp=(0.0, 1.0) if self.predicting else None, out_f32=True)
        return OrderedDict(enhanced=ops.to_nchw(y))


# ----------------------------------------------------------------------------------------------------------------
@MODELS.register(name="zero_dce_v", arch="zero_dce")
class ZeroDCEV(_GraphModel):
    """Zero-DCE-V (zero_dce_v.py:87-230): the curve net runs on the HSV V channel at ``down_size`` x ``down_size`` (nearest), the
    curve is applied there (``num_iters`` = 15 maps), a FastGuidedFilter(radius 1) carries the result to full resolution
    (bicubic interpolation of the linear coefficients), V is replaced, HSV -> RGB and the frame is divided by its maximum.
    Outputs ``image_v, image_v_lr, adjust, image_v_fixed_lr, image_v_fixed, enhanced`` (:206-213).  Keys ``e_conv{1..7}``."""

    arch = "zero_dce"
    tasks = [Task.LLIE]
    schemes = [Scheme.UNSUPERVISED, Scheme.ZERO_REFERENCE, Scheme.INSTANCE]

    def __init__(self, in_channels: int = 1, num_channels: int = 32, num_iters: int = 15, down_size: int = 256, weights: Any = None,
                 seed: int = None, *args, **kwargs):
        kwargs.pop("name", None)
        super().__init__(name="zero_dce_v", in_channels=3, weights=weights, *args, **kwargs)
        if isinstance(self.weights, dict):
            in_channels = self.weights.get("in_channels", in_channels)
            num_channels = self.weights.get("num_channels", num_channels)
            num_iters = self.weights.get("num_iters", num_iters)
            down_size = self.weights.get("down_size", down_size)
        if in_channels != 1:
            raise ValueError("zero_dce_v: the curve net works on the one-channel V image (in_channels = 1)")
        self.net_channels, self.num_channels, self.num_iters, self.down_size = in_channels, num_channels, num_iters, down_size
        c, o = num_channels, in_channels * num_iters
        for k, (ci, co) in enumerate([(in_channels, c), (c, c), (c, c), (c, c), (2 * c, c), (2 * c, c), (2 * c, o)], 1):
            self._params[f"e_conv{k}.weight"], self._params[f"e_conv{k}.bias"] = torch.zeros(co, ci, 3, 3), torch.zeros(co)
        self._finish_init(seed)

    def init_weights(self, seed: int = None):
        g = self._gen(seed)            # zero_dce_v.py:151-157: conv weight ~ N(0, 0.02), bias torch default
        for k in range(1, 8):
            self._init_conv(f"e_conv{k}", g, std=0.02)
        self._on_params_change()

    def forward(self, datapoint: dict, *args, **kwargs) -> dict:
        rgb = self._input(datapoint)
        hsv, v = ops.rgb_to_hsv(rgb)
        v_lr = ops.resize(v, (self.down_size, self.down_size), "nearest")
        x = v_lr if self.compute_dtype == "f32" else ops.cast(v_lr, self._tdt, 8)
        c = self._conv
        x1 = c(x, "e_conv1", act="relu")
        x2 = c(x1, "e_conv2", act="relu")
        x3 = c(x2, "e_conv3", act="relu")
        x4 = c(x3, "e_conv4", act="relu")
        x5 = c([x3, x4], "e_conv5", act="relu")
        x6 = c([x2, x5], "e_conv6", act="relu")
        xr = c([x1, x6], "e_conv7", act="tanh", out_f32=True)
        v_fixed_lr = ops.le_curve(v_lr, xr, self.num_iters)
        v_fixed = ops.fast_guided_filter(v_lr, v_fixed_lr, v, radius=1, eps=1e-8, clip=(0.0, 1.0))
        enhanced = ops.div_by_max(ops.hsv_to_rgb(hsv, v_fixed))
        out = OrderedDict(image_v=ops.to_nchw(v), image_v_lr=ops.to_nchw(v_lr), adjust=ops.to_nchw(xr), image_v_fixed_lr=ops.to_nchw(v_fixed_lr),
                          image_v_fixed=ops.to_nchw(v_fixed))
        out["enhanced"] = ops.to_nchw(enhanced)
        return out


# ----------------------------------------------------------------------------------------------------------------
@MODELS.register(name="gcenet", arch="gcenet")
class GCENet(_GraphModel):
    """GCENet (gcenet.py:338-502): ``EnhanceNet`` (:238-309) = seven DSConv2d + LeakyReLU(0.2) blocks (the last with tanh; ``norm=None`` as
    the model builds it, :387) on [image, depth, edge] with Zero-DCE's concat pattern -> ONE 3-channel curve map; edge =
    BoundaryAwarePrior(depth) (Sobel magnitude / max > ``dba_eps``); the curve runs ``num_iters`` times in its brightness-attention
    form (bam = (1 - V(median_blur(image, bam_ksize)))^bam_gamma, :478-484) and the result goes through GuidedFilter(gf_radius,
    gf_eps)(image, enhanced) -- whose windows are ``gf_radius`` pixels WIDE: the class passes its radius to BoxFilter as the kernel size (``ops.guided_filter_module``;
    pinned by tests/golden/guided_filter_seed161.npz).  ``datapoint`` needs ``depth`` (N,1,H,W) when ``use_depth``.  Keys ``en.e_conv{k}.conv.{dw,pw}_conv``."""

    arch = "gcenet"
    tasks = [Task.LLIE]
    schemes = [Scheme.UNSUPERVISED, Scheme.ZERO_REFERENCE]

    def __init__(self, in_channels: int = 3, num_channels: int = 32, num_iters: int = 15, dba_eps: float = 0.05, gf_radius: int = 3,
                 gf_eps: float = 1e-4, bam_gamma: float = 2.6, bam_ksize: int = 9, use_depth: bool = True, use_edge: bool = True,
                 weights: Any = None, seed: int = None, debug: bool = False, *args, **kwargs):
        kwargs.pop("name", None)
        super().__init__(name="gcenet", in_channels=in_channels, weights=weights, *args, **kwargs)
        self.num_channels, self.num_iters, self.dba_eps, self.gf_radius, self.gf_eps = num_channels, num_iters, dba_eps, gf_radius, gf_eps
        self.bam_gamma, self.bam_ksize, self.use_depth, self.use_edge, self.debug = bam_gamma, bam_ksize, use_depth, use_edge, debug
        ci0 = in_channels + (1 if use_depth else 0) + (1 if use_edge else 0)
        c = num_channels
        for k, (ci, co) in enumerate([(ci0, c), (c, c), (c, c), (c, c), (2 * c, c), (2 * c, c), (2 * c, 3)], 1):
            P, n = self._params, f"en.e_conv{k}.conv"
            P[f"{n}.dw_conv.weight"], P[f"{n}.dw_conv.bias"] = torch.zeros(ci, 1, 3, 3), torch.zeros(ci)
            P[f"{n}.pw_conv.weight"], P[f"{n}.pw_conv.bias"] = torch.zeros(co, ci, 1, 1), torch.zeros(co)
        self._finish_init(seed)

    def init_weights(self, seed: int = None):
        g = self._gen(seed)            # gcenet.py:268-281: dw / pw conv weights ~ N(0, 0.02)
        for k in range(1, 8):
            self._init_conv(f"en.e_conv{k}.conv.dw_conv", g, std=0.02)
            self._init_conv(f"en.e_conv{k}.conv.pw_conv", g, std=0.02)
        self._on_params_change()

    def _gray(self, x):
        if self._dev_cache is None:
            self._dev_cache = {}
        if "gray" not in self._dev_cache:      # kornia.color.rgb_to_grayscale: 0.299 R + 0.587 G + 0.114 B, as a 1x1 convolution
            self._dev_cache["gray"] = ops.ConvWeights(torch.tensor([0.299, 0.587, 0.114]).reshape(1, 3, 1, 1), None, self.device)
        return ops.conv2d([x], self._dev_cache["gray"], pad=0)

    def forward(self, datapoint: dict, *args, **kwargs) -> dict:
        image = self._input(datapoint)
        depth = datapoint.get("depth")
        if depth is not None:
            depth = ops.to_nhwc(depth.to(self.device, torch.float32))
            if depth.shape[1] == 3:
                depth = self._gray(depth)
        if self.use_depth and depth is None:
            raise ValueError("gcenet: `depth` must be given in the datapoint when use_depth=True")
        srcs, edge = [image], None
        if self.use_depth:
            srcs.append(depth)
        if self.use_edge:
            edge = ops.threshold_of_max(ops.sobel(depth if depth is not None else self._gray(image), 1e-6), self.dba_eps)
            srcs.append(edge)
        B = lambda k, ss, act="lrelu", f32=False: self._ds_block(f"en.e_conv{k}.conv", ss, act, f32, 0.2)
        if self.compute_dtype != "f32":
            # 16-bit path: EnhanceNet (the seven DSConv blocks) on the matrix cores, its input cat([image, depth, edge]) as one 8-channel 16-bit tensor;
            # the priors in front of it and everything behind the curve map (bam, LE-curve, guided filter) stay fp32 -- they are per-pixel streaming passes
            srcs = [ops.concat_cast(srcs, self._tdt, 8)]
        x1 = B(1, srcs); x2 = B(2, [x1]); x3 = B(3, [x2]); x4 = B(4, [x3])
        x5 = B(5, [x3, x4]); x6 = B(6, [x2, x5])
        adjust = B(7, [x1, x6], "tanh", True)
        bam = None
        if self.bam_gamma not in (None, 0.0):
            blurred = ops.median_blur(image, self.bam_ksize) if self.bam_ksize else image
            _, v = ops.rgb_to_hsv(blurred, want_hsv=False)
            bam = ops.pow_scalar(ops.affine(v, -1.0, 1.0), self.bam_gamma)                    # (1 - v)^gamma
        enhanced = ops.le_curve(image, adjust, self.num_iters, bam)
        enhanced = ops.guided_filter_module(image, enhanced, self.gf_radius, self.gf_eps)      # (windows gf_radius pixels WIDE, as the reference class filters)
        out = OrderedDict()
        if self.debug:
            out["adjust"] = ops.to_nchw(adjust)
            out["depth"] = None if depth is None else ops.to_nchw(depth)
            out["edge"] = None if edge is None else ops.to_nchw(edge)
            out["bam"] = None if bam is None else ops.to_nchw(bam)
        out["enhanced"] = ops.to_nchw(enhanced)
        return out


@MODELS.register(name="gcenet_zsn2n", arch="gcenet")
class GCENet_ZSN2N(GCENet):
    """gcenet.py:504-536: GCENet whose TRAINING loss adds the ZS-N2N pair-downsampling residual / consistency terms; ``forward`` is GCENet's, so checkpoints
    trained that way predict here unchanged.  The training step (``forward_loss``) is not built: GCENet has no training graph in this engine yet (DSConv
    adjoints, the brightness-attention curve and the guided filter backward -- DESIGN.md 8)."""

    def __init__(self, *args, **kwargs):
        kwargs.pop("name", None)
        super().__init__(*args, **kwargs)
        self.name = self.fullname = "gcenet_zsn2n"

    def forward_loss(self, datapoint: dict, *args, **kwargs) -> dict:
        raise capi.IEError("gcenet_zsn2n: the training step (gcenet.py:510-536) is not built in this engine; forward() / infer() are GCENet's")


@MODELS.register(name="gcenet_instance", arch="gcenet")
class GCENet_Instance(GCENet):
    """gcenet.py:539-600: GCENet fitted to ONE image at inference time (``infer`` runs ``epochs`` Adam steps on the zero-reference loss, then predicts).
    ``forward`` is GCENet's; the per-image fitting needs GCENet's training graph, which this engine does not have yet: ``infer(..., epochs=0)`` predicts
    with the loaded weights, any other ``epochs`` raises."""

    def __init__(self, *args, **kwargs):
        kwargs.pop("name", None)
        super().__init__(*args, **kwargs)
        self.name = self.fullname = "gcenet_instance"

    def infer(self, datapoint: dict, epochs: int = 300, lr: float = 0.00005, weight_decay: float = 0.00001, reset_weights: bool = True, *args, **kwargs) -> dict:
        if epochs:
            raise capi.IEError("gcenet_instance: per-image fitting (gcenet.py:552-600, `epochs` Adam steps on the zero-reference loss) needs GCENet's "
                               "training graph, which is not built in this engine; infer(..., epochs=0) predicts with the loaded weights")
        return super().infer(datapoint, *args, **kwargs)


# ----------------------------------------------------------------------------------------------------------------
@MODELS.register(name="retinexnet", arch="retinexnet")
class RetinexNet(_GraphModel):
    """DecomNet + RelightNet + ``S = R * I_delta`` (retinexnet/model.py:21-90, 199-238), single Decom pass.
    Checkpoint keys ``net1_*`` (decom) and ``net2_*`` (relight); the reference stores them in two files
    (``*_decom_*.tar`` / ``*_relight_*.tar``, model.py:455-465): pass ``weights={"decom": path, "relight": path}`` or
    one merged state_dict."""

    arch = "retinexnet"
    tasks = [Task.LLIE]
    schemes = [Scheme.SUPERVISED]

    def __init__(self, channels: int = 64, weights: Any = None, seed: int = None, *args, **kwargs):
        kwargs.pop("name", None)
        two = isinstance(weights, dict) and "decom" in weights and "relight" in weights
        super().__init__(name="retinexnet", in_channels=3, weights=None if two else weights, *args, **kwargs)
        self.predicting = True
        c = self.channels = channels
        P = self._params
        P["net1_conv0.weight"], P["net1_conv0.bias"] = torch.zeros(c, 4, 9, 9), torch.zeros(c)
        for i in (0, 2, 4, 6, 8):
            P[f"net1_convs.{i}.weight"], P[f"net1_convs.{i}.bias"] = torch.zeros(c, c, 3, 3), torch.zeros(c)
        P["net1_recon.weight"], P["net1_recon.bias"] = torch.zeros(4, c, 3, 3), torch.zeros(4)
        P["net2_conv0_1.weight"], P["net2_conv0_1.bias"] = torch.zeros(c, 4, 3, 3), torch.zeros(c)
        for i in (1, 2, 3):
            P[f"net2_conv1_{i}.weight"], P[f"net2_conv1_{i}.bias"] = torch.zeros(c, c, 3, 3), torch.zeros(c)
            P[f"net2_deconv1_{i}.weight"], P[f"net2_deconv1_{i}.bias"] = torch.zeros(c, 2 * c, 3, 3), torch.zeros(c)
        P["net2_fusion.weight"], P["net2_fusion.bias"] = torch.zeros(c, 3 * c, 1, 1), torch.zeros(c)
        P["net2_output.weight"], P["net2_output.bias"] = torch.zeros(1, c, 3, 3), torch.zeros(1)
        if two:
            from .weights import read_state_dict
            sd = dict(read_state_dict(weights["decom"]) if not isinstance(weights["decom"], dict) else weights["decom"])
            sd.update(read_state_dict(weights["relight"]) if not isinstance(weights["relight"], dict) else weights["relight"])
            self.load_state_dict(sd)
        else:
            self._finish_init(seed)

    def map_key(self, k: str) -> str:
        for pre in ("DecomNet.", "RelightNet."):
            if k.startswith(pre):
                k = k[len(pre):]
        return k

    def init_weights(self, seed: int = None):
        g = self._gen(seed)
        for k in list(self._params):
            if k.endswith(".weight"):
                self._init_conv(k[:-7], g)
        self._on_params_change()

    def decom(self, x, copy16=False, copy16_channels=8):
        """x: fp32 NHWC image (fp32 path) or DecomNet's 16-bit 8-channel input ``[max_c, R, G, B, 0..]`` (16-bit path, ops.retinex_decom_input).
        Returns DecomNet's sigmoid output, fp32 NHWC: channels 0-2 = R, 3 = L; with ``copy16`` also its 16-bit copy padded to 8 channels."""
        R, c = "replicate", self._conv
        if self.compute_dtype == "f32":
            xin = ops.max_concat(x)                                                # [max over channels, x], model.py:45-46
        else:
            xin = x
        if xin.shape[1] == 4 and self.compute_dtype != "f32":      # the 4-channel 16-bit input: the 9x9 strip kernel (ie_conv_first9.hip)
            key = ("conv9x9c4", "net1_conv0")
            if self._dev_cache is None:
                self._dev_cache = {}
            if key not in self._dev_cache:
                self._dev_cache[key] = ops.Conv9x9C4Weights(self._params["net1_conv0.weight"], self._params["net1_conv0.bias"], self._tdt, self.device)
            f = ops.conv9x9_c4(xin, self._dev_cache[key])
        else:
            f = c(xin, "net1_conv0", pad=4, pad_mode=R)
        layers = [0, 2, 4, 6, 8]
        while layers:
            i = layers.pop(0)
            if self.compute_dtype != "f32" and layers:      # two 64 -> 64 layers per launch, the one between them in LDS only (ie_conv_pair.hip)
                mwa, mwb = self._mw_plain((self.channels,), f"net1_convs.{i}"), self._mw_plain((self.channels,), f"net1_convs.{layers[0]}")
                if ops.mconv2d_pair_applicable(f, mwa, mwb, pad_mode=R):
                    f = ops.mconv2d_pair(f, mwa, mwb, pad_mode=R)
                    layers.pop(0)
                    continue
            if self.compute_dtype != "f32" and not layers:  # the last 64 -> 64 layer and net1_recon (64 -> 4, sigmoid, fp32 out) in one launch
                mwa, mwb = self._mw_plain((self.channels,), f"net1_convs.{i}"), self._mw_plain((self.channels,), "net1_recon")
                if ops.mconv2d_pair_thin_applicable(f, mwa, mwb, pad_mode=R, act_b="sigmoid"):
                    return ops.mconv2d_pair_thin(f, mwa, mwb, pad_mode=R, act_b="sigmoid", copy16=copy16, copy16_channels=copy16_channels)
            f = c(f, f"net1_convs.{i}", pad_mode=R, act="relu")
        # 16-bit path: R and L leave in fp32 (16 bytes per pixel either way): S = R * I_delta multiplies by R itself, not by its 16-bit
        # rounding -- that rounding alone was 7e-4 of the 1e-3 bound (tools/retinex_attrib.py)
        rl = c(f, "net1_recon", pad_mode=R, act="sigmoid", out_f32=self.compute_dtype != "f32")
        return (rl, ops.cast(rl, self._tdt, copy16_channels)) if copy16 else rl

    def relight(self, rl):
        """rl: cat(R, L) == the 4-channel sigmoid output of DecomNet (model.py:72)."""
        R, c = "replicate", self._conv
        if rl.shape[1] == 4 and self.compute_dtype != "f32":        # (R, L) as a 4-channel 16-bit tensor: the first-layer strip kernel at 3x3 (ie_conv_first9.hip)
            key = ("conv3x3c4", "net2_conv0_1")
            if self._dev_cache is None:
                self._dev_cache = {}
            if key not in self._dev_cache:
                self._dev_cache[key] = ops.Conv9x9C4Weights(self._params["net2_conv0_1.weight"], self._params["net2_conv0_1.bias"], self._tdt, self.device)
            out0 = ops.conv9x9_c4(rl, self._dev_cache[key])
        else:
            out0 = c(rl, "net2_conv0_1", pad_mode=R)
        out1 = c(out0, "net2_conv1_1", stride=2, pad_mode=R, act="relu")
        out2 = c(out1, "net2_conv1_2", stride=2, pad_mode=R, act="relu")
        out3 = c(out2, "net2_conv1_3", stride=2, pad_mode=R, act="relu")
        # F.interpolate(x, size=...) (nearest) in front of a conv: its own pass on the fp32 path; on the 16-bit path the conv kernels read
        # the small tensor through the up-sampling (ops.mconv2d: a source smaller than the largest one)
        if self.compute_dtype == "f32":
            rs = lambda t, ref: ops.resize(t, (ref.shape[2], ref.shape[3]), "nearest")
        else:
            rs = lambda t, ref: t
        d1 = c([rs(out3, out2), out2], "net2_deconv1_1", pad_mode=R, act="relu")
        d2 = c([rs(d1, out1), out1], "net2_deconv1_2", pad_mode=R, act="relu")
        d3 = c([rs(d2, out0), out0], "net2_deconv1_3", pad_mode=R, act="relu")
        if self.compute_dtype != "f32":
            # net2_fusion (1x1, padding=1 replicate, NO activation) followed by net2_output (3x3 valid) is one linear map: a 1x1 conv
            # commutes with replicate padding, so the pair equals ONE 3x3 conv 192 -> 1 with replicate padding 1 and weights
            # W[c, ky, kx] = sum_m W_out[m, ky, kx] * W_fus[m, c] (composed on the host in fp64).  The 64-channel `fus` tensor -- whose 16-bit
            # rounding carried 5.6e-4 of the bound -- never exists, and the pair costs 1,728 instead of 12,864 MACs per pixel.
            key = ("mconv-composed", "net2_fusion*net2_output")
            if self._dev_cache is None:
                self._dev_cache = {}
            if key not in self._dev_cache:
                wf, bf = self._params["net2_fusion.weight"].double()[:, :, 0, 0], self._params["net2_fusion.bias"].double()      # (64, 192), (64)
                wo, bo = self._params["net2_output.weight"].double(), self._params["net2_output.bias"].double()                 # (1, 64, 3, 3), (1)
                w = torch.einsum("omyx,mc->ocyx", wo, wf).float().contiguous()
                b = (bo + torch.einsum("omyx,m->o", wo, bf)).float()
                cc = self.channels
                self._dev_cache[key] = ops.MConvWeights(w, b, (cc, cc, cc), self._tdt, self.device)
            return ops.mconv2d([d1, d2, d3], self._dev_cache[key], stride=1, pad=1, pad_mode=R, out_f32=True)
        fus = c([rs(d1, rl), rs(d2, rl), d3], "net2_fusion",
                pad=1, pad_mode=R)                                                 # 1x1 with replicate padding 1: (H+2, W+2)
        return c(fus, "net2_output", pad=0, out_f32=True)                          # 3x3 valid -> (H, W), fp32

    # ---- training (model.py:199-238 forward with its losses; model.py:324-424 train loop) -------------------------------------------------
    # The reference trains in two runs, each with its own fresh Adam: phase "decom" steps DecomNet on loss_Decom, phase "relight" steps
    # RelightNet on loss_Relight.  Here: set ``train_phase`` and build one Trainer per phase (the other net's gradients and Adam moments
    # stay zero, so it does not move); ``Trainer.sync_to_model()`` hands the trained weights to the next phase / to inference.
    train_phase = "decom"
    supports_mixed_training = True        # round 3: replicate padding and stride 2 on the 16-bit graph (ie_pad_replicate_nhwc, ie_zero_insert2_nhwc)

    def forward_train_loss(self, datapoint: dict, store, tape) -> dict:
        from . import train as T
        from .losses import LossResult
        if self.train_phase not in ("decom", "relight"):
            raise ValueError(f"`train_phase` must be 'decom' or 'relight', but got {self.train_phase!r}.")
        mixed = store.compute_dtype != "f32"          # 16-bit activations / gradients between fp32 edges (xin, (R, L), I_delta, the losses): see the `mixed` branches
        tdt = ops.TORCH_DT[store.compute_dtype]
        if datapoint.get("ref_image") is None:
            raise ValueError("RetinexNet's training forward needs the normal-light frame as `ref_image` (model.py:199: input_high).")
        low = self._input(datapoint)
        high = datapoint["ref_image"]
        high = ops.to_nhwc((high[None] if high.dim() == 3 else high).to(self.device, torch.float32))
        if tuple(high.shape) != tuple(low.shape):
            raise ValueError(f"`image` {tuple(low.shape)} and `ref_image` {tuple(high.shape)} must have one shape.")
        n, _, h, w = low.shape
        decom_phase = self.train_phase == "decom"
        idle = T.Tape(recording=False)                                                  # the other net runs forward only: its tape keeps no closures
        tape_d, tape_r = (tape, idle) if decom_phase else (idle, tape)
        store.grads.zero_()                                                             # the idle net's gradients must read zero

        def conv(tp, name, srcs, pad, act="none", stride=1, cout_pad=None):
            srcs = [T.t_pad_replicate(tp, t, pad) for t in (srcs if isinstance(srcs, (list, tuple)) else [srcs])]
            return T.t_conv(tp, srcs, store.conv(name), *store.conv_grads(name), stride=stride, pad=0, act=act, cout_pad=cout_pad if mixed else None)

        both = ops.empty_nhwc(2 * n, 3, h, w, low.device)                               # DecomNet on low and high in one batch (shared weights)
        both[:n].copy_(low); both[n:].copy_(high)
        xin = ops.max_concat(both)
        tape_d.no_grad(xin)
        if mixed:
            xin = T.t_cast(tape_d, xin, tdt, channels=16)                               # 4 channels carried as 16: the 9x9 first layer runs on the matrix cores too
        f = conv(tape_d, "net1_conv0", xin, 4)
        for i in (0, 2, 4, 6, 8):
            f = conv(tape_d, f"net1_convs.{i}", f, 1, "relu")
        rl = conv(tape_d, "net1_recon", f, 1, "sigmoid", cout_pad=16)                   # (2N, 4, H, W): channels 0-2 = R, 3 = I (mixed: carried as 16)
        rl_low = rl[:n]
        if mixed:
            rl_low._ie_real = 4
            rl = T.t_cast(tape_d, rl, torch.float32, channels=4)                        # the losses read (and seed) the fp32 copy
        tape_r.no_grad(rl_low)
        rs = lambda t, ref: T.t_resize(tape_r, t, (ref.shape[2], ref.shape[3]), "nearest")
        out0 = conv(tape_r, "net2_conv0_1", rl_low, 1)
        out1 = conv(tape_r, "net2_conv1_1", out0, 1, "relu", 2)
        out2 = conv(tape_r, "net2_conv1_2", out1, 1, "relu", 2)
        out3 = conv(tape_r, "net2_conv1_3", out2, 1, "relu", 2)
        d1 = conv(tape_r, "net2_deconv1_1", [rs(out3, out2), out2], 1, "relu")
        d2 = conv(tape_r, "net2_deconv1_2", [rs(d1, out1), out1], 1, "relu")
        d3 = conv(tape_r, "net2_deconv1_3", [rs(d2, out0), out0], 1, "relu")
        fus = conv(tape_r, "net2_fusion", [rs(d1, rl_low), rs(d2, rl_low), d3], 1)     # 1x1 with replicate padding 1: (H+2, W+2)
        i_delta = conv(tape_r, "net2_output", fus, 0, cout_pad=16)                      # 3x3 valid -> (N, 1, H, W) (mixed: carried as 16: an 8-channel tensor would be a "thin" source of its data-gradient conv)
        if mixed:
            i_delta = T.t_cast(tape_r, i_delta, torch.float32, channels=1)

        # ---- losses (model.py:216-238) and their gradients with respect to rl (decom phase) / i_delta (relight phase) ----------------
        L = capi.lib()
        import ctypes as C
        acc = torch.zeros(2, device=low.device, dtype=torch.float64)                    # {loss_Decom, loss_Relight}
        d_rl = torch.zeros_like(rl) if decom_phase else None
        d_delta = None if decom_phase else torch.zeros_like(i_delta)
        ws = torch.empty(int(L.ie_retinex_smooth_ws_floats(n, h, w)), device=low.device, dtype=torch.float32)
        off = lambda t, k: None if t is None else C.c_void_p(t.data_ptr() + 4 * k)
        lo_, hi_ = (lambda t: None if t is None else t[:n]), (lambda t: None if t is None else t[n:])
        R_l, I_l, R_h, I_h = off(rl[:n], 0), off(rl[:n], 3), off(rl[n:], 0), off(rl[n:], 3)
        dR_l, dI_l, dR_h, dI_h = off(lo_(d_rl), 0), off(lo_(d_rl), 3), off(hi_(d_rl), 0), off(hi_(d_rl), 3)
        a_dec, a_rel = off(acc, 0), C.c_void_p(acc.data_ptr() + 8)
        st = capi.current_stream
        with torch.cuda.device(low.device):
            recon = lambda R, rs_, I, is_, Tg, wgt, dR, dI, a: capi.check(
                L.ie_retinex_recon_f32(R, rs_, I, is_, capi.ptr(Tg), wgt, dR, dI, a, n, h, w, st()), "ie_retinex_recon_f32")
            smooth = lambda I, is_, R, wgt, dI, dR, a: capi.check(
                L.ie_retinex_smooth_f32(I, is_, R, 4, wgt, dI, dR, a, capi.ptr(ws), n, h, w, st()), "ie_retinex_smooth_f32")
            recon(R_l, 4, I_l, 4, low, 1.0, dR_l, dI_l, a_dec)                          # recon_loss_low
            recon(R_h, 4, I_h, 4, high, 1.0, dR_h, dI_h, a_dec)                         # recon_loss_high
            recon(R_h, 4, I_l, 4, low, 0.001, dR_h, dI_l, a_dec)                        # recon_loss_mutal_low  = L1(R_high * I_low,  input_low)
            recon(R_l, 4, I_h, 4, high, 0.001, dR_l, dI_h, a_dec)                       # recon_loss_mutal_high = L1(R_low * I_high, input_high)
            smooth(I_l, 4, R_l, 0.1, dI_l, dR_l, a_dec)                                 # Ismooth_loss_low
            smooth(I_h, 4, R_h, 0.1, dI_h, dR_h, a_dec)                                 # Ismooth_loss_high
            capi.check(L.ie_l1_detached_f32(R_l, R_h, 4, 3, 0.01, dR_l, a_dec, n * h * w, st()), "ie_l1_detached_f32")    # equal_R_loss
            recon(R_l, 4, capi.ptr(i_delta), 1, high, 1.0, None, capi.ptr(d_delta), a_rel)                               # relight_loss
            smooth(capi.ptr(i_delta), 1, R_l, 3.0, capi.ptr(d_delta), None, a_rel)                                        # 3 * Ismooth_loss_delta
        rl_low32 = rl[:n]
        rl_nchw = ops.to_nchw(rl_low32)
        outputs = OrderedDict()
        outputs["reflectance"], outputs["illumination"] = rl_nchw[:, 0:3], rl_nchw[:, 3:4]
        outputs["illumination_delta"] = ops.to_nchw(i_delta)
        outputs["enhanced"] = ops.to_nchw(ops.mul_prefix(rl_low32, i_delta))
        outputs["loss_decom"], outputs["loss_relight"] = acc[0], acc[1]

        def seed(tape_, scale=1.0):
            g = d_rl if decom_phase else d_delta
            tape_.seed(rl if decom_phase else i_delta, g if scale == 1.0 else ops.affine(g, scale))
        outputs["loss"] = LossResult(lambda: float(acc[0 if decom_phase else 1].item()), None, seed)
        return outputs

    def forward(self, datapoint: dict, *args, **kwargs) -> dict:
        out = OrderedDict()
        if self.compute_dtype != "f32":
            # 16-bit path: the image enters as DecomNet's 8-channel 16-bit input in one pass, DecomNet's last launch writes (R, L) in fp32 AND as
            # RelightNet's 16-bit input, and the four NCHW outputs leave in one pass (ie_retinex_decom_input / ie_retinex_outputs_f32)
            self.assert_datapoint(datapoint)
            self._require_cuda()
            x = datapoint.get("image")
            x = x[None] if x.dim() == 3 else x
            if x.dim() != 4 or x.shape[1] != self.in_channels:
                raise ValueError(f"`image` must be (N,{self.in_channels},H,W), but got {tuple(x.shape)}.")
            c4 = self.channels == 64 and os.environ.get("IE_RETINEX_CONV9", "1") != "0"      # (IE_RETINEX_CONV9=0: A/B switch back to the tiled thin-source kernel)
            rl, rl16 = self.decom(ops.retinex_decom_input(x.to(self.device, torch.float32), self._tdt, 4 if c4 else 8), copy16=True, copy16_channels=4 if c4 else 8)
            i_delta = self.relight(rl16)
            out["reflectance"], out["illumination"], s = ops.retinex_outputs(rl, i_delta)
            out["illumination_delta"] = i_delta.contiguous()                        # (N,1,H,W): NHWC and NCHW are the same bytes
            out["enhanced"] = s
            return out
        x = self._input(datapoint)
        rl = self.decom(x)
        i_delta = self.relight(rl)
        rl_nchw = ops.to_nchw(rl)
        s = ops.mul_prefix(rl, i_delta)
        out["reflectance"] = rl_nchw[:, 0:3]
        out["illumination"] = rl_nchw[:, 3:4]
        out["illumination_delta"] = ops.to_nchw(i_delta)
        out["enhanced"] = ops.to_nchw(s)
        return out


# ----------------------------------------------------------------------------------------------------------------
@MODELS.register(name="hinet_re", arch="hinet")
class HINet_RE(_GraphModel):
    """HINet (hinet.py:204-362): two U-Net stages of ``UNetConvBlock`` (hinet.py:34-87: Conv3x3 -> InstanceNorm on the
    first half of the channels -> LeakyReLU -> Conv3x3 -> LeakyReLU, + Conv1x1 identity, + cross-stage feature fusion,
    Conv4x4 stride-2 down-sampling), ``UNetUpBlock`` (ConvTranspose 2x2 + cat + block, :90-119), supervised attention
    between the stages (:122-198).  Outputs ``stage1`` and ``enhanced`` (last).

    The reference's ``if enc and dec:`` (hinet.py:80) raises on multi-element tensors, so stage 2 cannot run as shipped;
    the intended test (``enc is not None and dec is not None``, as in the original HINet) is implemented.
    Checkpoint keys are the reference module names (``down_path_1.0.conv_1.weight`` ... ``sam12.conv3.bias``, ``last``)."""

    arch = "hinet"
    tasks = [Task.DEBLUR, Task.DENOISE, Task.DERAIN, Task.DESNOW, Task.LES]
    schemes = [Scheme.SUPERVISED]

    def __init__(self, in_channels: int = 3, num_channels: int = 64, depth: int = 5, relu_slope: float = 0.2, in_pos_left: int = 0,
                 in_pos_right: int = 4, weights: Any = None, seed: int = None, *args, **kwargs):
        kwargs.pop("name", None)
        super().__init__(name="hinet_re", in_channels=in_channels, weights=weights, *args, **kwargs)
        if isinstance(self.weights, dict):
            in_channels = self.weights.get("in_channels", in_channels)
            num_channels = self.weights.get("num_channels", num_channels)
            depth = self.weights.get("depth", depth)
            relu_slope = self.weights.get("relu_slope", relu_slope)
            in_pos_left = self.weights.get("in_pos_left", in_pos_left)
            in_pos_right = self.weights.get("in_pos_right", in_pos_right)
        self.in_channels, self.num_channels, self.depth, self.relu_slope = in_channels, num_channels, depth, relu_slope
        self.in_pos_left, self.in_pos_right = in_pos_left, in_pos_right
        self.predicting = True
        P = self._params

        def conv(name, ci, co, k, bias=True):
            P[f"{name}.weight"] = torch.zeros(co, ci, k, k)
            if bias:
                P[f"{name}.bias"] = torch.zeros(co)

        def block(name, ci, co, down, csff, hin):
            conv(f"{name}.identity", ci, co, 1); conv(f"{name}.conv_1", ci, co, 3); conv(f"{name}.conv_2", co, co, 3)
            if down and csff:
                conv(f"{name}.csff_enc", co, co, 3); conv(f"{name}.csff_dec", co, co, 3)
            if hin:
                P[f"{name}.norm.weight"], P[f"{name}.norm.bias"] = torch.ones(co // 2), torch.zeros(co // 2)
            if down:
                conv(f"{name}.downsample", co, co, 4, bias=False)

        c = num_channels
        conv("conv_01", in_channels, c, 3); conv("conv_02", in_channels, c, 3)
        self._hin, prev = [], c
        for i in range(depth):
            hin, down = in_pos_left <= i <= in_pos_right, (i + 1) < depth
            self._hin.append(hin)
            block(f"down_path_1.{i}", prev, (2 ** i) * c, down, False, hin)
            block(f"down_path_2.{i}", prev, (2 ** i) * c, down, down, hin)
            prev = (2 ** i) * c
        for j, i in enumerate(reversed(range(depth - 1))):
            for s in (1, 2):
                P[f"up_path_{s}.{j}.up.weight"], P[f"up_path_{s}.{j}.up.bias"] = torch.zeros(prev, (2 ** i) * c, 2, 2), torch.zeros((2 ** i) * c)
                block(f"up_path_{s}.{j}.conv_block", prev, (2 ** i) * c, False, False, False)
                conv(f"skip_conv_{s}.{j}", (2 ** i) * c, (2 ** i) * c, 3)
            prev = (2 ** i) * c
        conv("sam12.conv1", prev, prev, 3); conv("sam12.conv2", prev, 3, 3); conv("sam12.conv3", 3, prev, 3)
        conv("cat12", prev * 2, prev, 1); conv("last", prev, in_channels, 3)
        self._finish_init(seed)

    def init_weights(self, seed: int = None):
        # hinet.py:301-308 only touches modules that have a ``.conv`` attribute (none of these), i.e. torch's default
        # Conv2d / ConvTranspose2d init stays: U(+-1/sqrt(fan_in)); ConvTranspose2d's fan_in is Cout*k*k (weight dim 1).
        g = self._gen(seed)
        for k in list(self._params):
            if k.endswith(".weight") and self._params[k].dim() == 4:
                self._init_conv(k[:-7], g)
        self._on_params_change()

    def _block(self, name, srcs, hin, down, enc=None, dec=None):
        s, c = self.relu_slope, self._conv
        idn = c(srcs, f"{name}.identity", pad=0)
        if hin:
            y = c(srcs, f"{name}.conv_1")
            half = y.shape[1] // 2
            y = ops.instance_norm(y, self._vec(f"{name}.norm.weight"), self._vec(f"{name}.norm.bias"), 1e-5, "lrelu", s, 0, half)
        else:
            y = c(srcs, f"{name}.conv_1", act="lrelu", slope=s)
        y = c(y, f"{name}.conv_2", act="lrelu", slope=s, residual=idn)
        if enc is not None and dec is not None:
            y = c(enc, f"{name}.csff_enc", residual=y)
            y = c(dec, f"{name}.csff_dec", residual=y)
        if down:
            return c(y, f"{name}.downsample", stride=2, pad=1), y
        return y

    def conv_transpose_keys(self):
        return [k for k in self._params if k.endswith(".up.weight")]

    supports_mixed_training = True

    def forward_train(self, x, store, tape):
        """Training graph of hinet.py:34-198, 322-362 on the taped ops; returns ``stage1`` and ``enhanced`` (NHWC fp32), whose losses the
        trainer sums (hinet.py:311-320).  16-bit graph (``Trainer(compute_dtype="bf16" | "f16")``): every stride-1 convolution, the
        half instance norms, adds and the attention product run on 16-bit tensors (RGB carried as 16 channels), since round 3 the Conv4x4
        stride-2 layers too (data gradient = stride-1 conv of the zero-inserted output gradient, train.t_conv16); ConvTranspose 2x2 still takes
        an fp32 detour (cast, fp32 kernels for forward / data / weight gradient, cast back): a few per cent of the FLOPs."""
        from . import train as T
        m = 2 ** (self.depth - 1)
        if x.shape[2] % m or x.shape[3] % m:
            raise ValueError(f"HINet needs H and W divisible by {m} (got {x.shape[2]}x{x.shape[3]}).")
        s_ = self.relu_slope
        g = store.grads
        mixed = store.compute_dtype != "f32"
        tdt = ops.TORCH_DT[store.compute_dtype]
        if mixed:
            if self.num_channels % 16:
                raise capi.IEError("hinet: the 16-bit graph needs num_channels divisible by 16 (half instance norm on 8-channel groups)")
            x = T.t_cast(tape, x, tdt, channels=16)

        def conv(name, srcs, pad=1, act="none", stride=1, cout_pad=None):
            if mixed and stride != 1 and os.environ.get("IE_HINET_S2_FP32") == "1":      # A/B: the round-2 fp32 detour of the Conv4x4 stride-2 layers
                srcs = [T.t_cast(tape, t, torch.float32) for t in srcs]
                y = T.t_conv(tape, srcs, store.conv(name), *store.conv_grads(name), stride=stride, pad=pad, act=act, slope=s_)
                return T.t_cast(tape, y, tdt)
            return T.t_conv(tape, srcs, store.conv(name), *store.conv_grads(name), stride=stride, pad=pad, act=act, slope=s_,
                            cout_pad=cout_pad if mixed else None)

        def up2x2(t, name):
            if mixed and os.environ.get("IE_CONVT_FP32") == "1":      # A/B: the fp32 detour of rounds 2 - 3
                return T.t_cast(tape, T.t_conv_transpose2x2(tape, T.t_cast(tape, t, torch.float32), store, name), tdt)
            return T.t_conv_transpose2x2(tape, t, store, name)

        def block(name, srcs, hin, down, enc=None, dec=None):
            idn = conv(f"{name}.identity", srcs, pad=0)
            y = conv(f"{name}.conv_1", srcs, act="none" if hin else "lrelu")
            if hin:
                half = y.shape[1] // 2
                y = T.t_norm(tape, y, store.view(f"{name}.norm.weight"), store.view(f"{name}.norm.bias"), store.view(f"{name}.norm.weight", g),
                             store.view(f"{name}.norm.bias", g), per_sample=True, eps=1e-5, act="lrelu", slope=s_, c_begin=0, c_count=half)
            y = T.t_add(tape, conv(f"{name}.conv_2", [y], act="lrelu"), idn)
            if enc is not None and dec is not None:
                y = T.t_add(tape, T.t_add(tape, y, conv(f"{name}.csff_enc", [enc])), conv(f"{name}.csff_dec", [dec]))
            if down:
                return conv(f"{name}.downsample", [y], stride=2, pad=1), y
            return y

        def stage(s, t, encs=None, decs=None):
            ups, outs = [], []
            for i in range(self.depth):
                name = f"down_path_{s}.{i}"
                if i + 1 < self.depth:
                    e, dd = (encs[i], decs[-i - 1]) if encs is not None else (None, None)
                    t, t_up = block(name, [t], self._hin[i], True, e, dd)
                    ups.append(t_up)
                else:
                    t = block(name, [t], self._hin[i], False)
            for j in range(self.depth - 1):
                bridge = conv(f"skip_conv_{s}.{j}", [ups[-j - 1]])
                t = block(f"up_path_{s}.{j}.conv_block", [up2x2(t, f"up_path_{s}.{j}.up"), bridge], False, False)
                outs.append(t)
            return t, ups, outs

        x1, encs, decs = stage(1, conv("conv_01", [x]))
        f1 = conv("sam12.conv1", [x1])
        img = T.t_add(tape, conv("sam12.conv2", [x1], cout_pad=16), x)
        att = conv("sam12.conv3", [img], act="sigmoid")
        sam = T.t_add(tape, T.t_mul(tape, f1, att), x1)
        x2 = conv("cat12", [conv("conv_02", [x]), sam], pad=0)
        x2, _, _ = stage(2, x2, encs, decs)
        y2 = T.t_add(tape, conv("last", [x2], cout_pad=16), x)
        if mixed:                                      # the losses run in fp32 on the 3-channel predictions
            img, y2 = T.t_cast(tape, img, torch.float32, channels=3), T.t_cast(tape, y2, torch.float32, channels=3)
        return OrderedDict(stage1=img, enhanced=y2)

    def _stage(self, s, x, encs=None, decs=None):
        d = self.depth
        ups, outs = [], []
        for i in range(d):
            name = f"down_path_{s}.{i}"
            if i + 1 < d:
                e, dd = (encs[i], decs[-i - 1]) if encs is not None else (None, None)
                x, x_up = self._block(name, [x], self._hin[i], True, e, dd)
                ups.append(x_up)
            else:
                x = self._block(name, [x], self._hin[i], False)
        for j in range(d - 1):
            bridge = self._conv(ups[-j - 1], f"skip_conv_{s}.{j}")
            x = self._block(f"up_path_{s}.{j}.conv_block", [self._conv_transpose2x2(x, f"up_path_{s}.{j}.up"), bridge], False, False)
            outs.append(x)
        return x, ups, outs

    def forward(self, datapoint: dict, *args, **kwargs) -> dict:
        x = self._input(datapoint)
        m = 2 ** (self.depth - 1)
        if x.shape[2] % m or x.shape[3] % m:
            raise ValueError(f"HINet needs H and W divisible by {m} (got {x.shape[2]}x{x.shape[3]}); use infer(), which resizes "
                             f"to a multiple of 32 like the reference (vision/model.py:106).")
        f32 = self.compute_dtype == "f32"
        if not f32:
            if self.num_channels % 16:
                raise capi.IEError("hinet_re: the 16-bit path needs num_channels to be a multiple of 16")
            x = ops.cast(x, self._tdt, 8)                                   # RGB carried as 8 channels (zeros appended)
        cp = None if f32 else 8
        c = self._conv
        x1, encs, decs = self._stage(1, c(x, "conv_01"))
        # supervised attention (hinet.py:192-197)
        f1 = c(x1, "sam12.conv1")
        img = c(x1, "sam12.conv2", residual=x, cout_pad=cp)
        att = c(img, "sam12.conv3", act="sigmoid")
        sam = ops.add(ops.mul(f1, att), x1)
        x2 = c([c(x, "conv_02"), sam], "cat12", pad=0)
        x2, _, _ = self._stage(2, x2, encs, decs)
        y2 = c(x2, "last", residual=x, cout_pad=cp)
        if not f32:
            img, y2 = ops.cast(img, torch.float32, 3), ops.cast(y2, torch.float32, 3)
        return OrderedDict(stage1=ops.to_nchw(img), enhanced=ops.to_nchw(y2))
