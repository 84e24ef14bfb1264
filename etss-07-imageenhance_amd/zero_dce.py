"""Zero-DCE on the HIP plan: host-side mirror of ``ZeroDCE_RE``
(src/mon/vision/enhance/llie/zero_dce/zero_dce.py:86-191; original network
src/mon_extra/vision/enhance/llie/zero_dce/model.py:11-50).

Same registry entry (``name="zero_dce_re", arch="zero_dce"``), constructor hyper-parameters,
state_dict key names (``e_conv{1..7}.{weight,bias}``; ``loss.*`` ignored) and output dict
(``adjust_0..7``, ``adjust``, ``enhanced`` inserted last) as the reference.  The forward itself is
one call into ``ie_zdce_forward`` (include/ie_hip.h).
"""
from __future__ import annotations

import ctypes as C
from collections import OrderedDict
from typing import Any

import torch

from . import capi
from .globals_ import Scheme, Task
from .model import ImageEnhancementModel
from .registry import MODELS

_LAYERS = tuple(f"e_conv{k}" for k in range(1, 8))


@MODELS.register(name="zero_dce_re", arch="zero_dce")
class ZeroDCE_RE(ImageEnhancementModel):
    """Args mirror zero_dce.py:107-131.  Extra (engine) arguments:
        compute_dtype: "f16" | "bf16" | "bf16_pure" (MFMA path, fp32 accumulate) or "f32" (VALU parity mode).
            "bf16" holds the tensors that carry the rounding error (x1, x2, x6, the partial sum P and the weights
            contracted with them) in fp16 and x3/x4/x5 in bf16: 6e-4 of the fp32 result at the same bytes and MFMA
            rate; "bf16_pure" (= variant | capi.VAR_PURE_BF16) keeps every tensor and weight in bf16: 2e-3
            (tools/bf16_attrib.py).
        variant: kernel variant bitmask (capi.VAR_*), 0 = default.
    """

    arch: str = "zero_dce"
    tasks: list[Task] = [Task.LLIE]
    schemes: list[Scheme] = [Scheme.UNSUPERVISED, Scheme.ZERO_REFERENCE]
    zoo: dict = {}

    def __init__(self, in_channels: int = 3, num_channels: int = 32, num_iters: int = 8, weights: Any = None,
                 compute_dtype: str = "f16", variant: int = 0, seed: int = None, *args, **kwargs):
        kwargs.pop("name", None)
        super().__init__(name="zero_dce_re", in_channels=in_channels, weights=weights, *args, **kwargs)
        if isinstance(self.weights, dict):   # zero_dce.py:122-126
            in_channels = self.weights.get("in_channels", in_channels)
            num_channels = self.weights.get("num_channels", num_channels)
            num_iters = self.weights.get("num_iters", num_iters)
        self.in_channels = in_channels or self.in_channels
        self.num_channels = num_channels
        self.num_iters = num_iters
        self.out_channels = self.in_channels * self.num_iters
        if compute_dtype == "bf16_pure":
            compute_dtype, variant = "bf16", int(variant) | capi.VAR_PURE_BF16
        if compute_dtype not in capi.DTYPES:
            raise ValueError(f"`compute_dtype` must be one of {sorted(capi.DTYPES) + ['bf16_pure']}, but got {compute_dtype}.")
        self.compute_dtype = compute_dtype
        self.variant = int(variant)
        self._plan = None
        self._plan_device = None
        self._plan_dirty = True
        self._tuning = (0, 0, 0)
        c, o = self.num_channels, self.out_channels
        shapes = [(c, self.in_channels), (c, c), (c, c), (c, c), (c, 2 * c), (c, 2 * c), (o, 2 * c)]
        for name, (co, ci) in zip(_LAYERS, shapes):
            self._params[f"{name}.weight"] = torch.zeros(co, ci, 3, 3)
            self._params[f"{name}.bias"] = torch.zeros(co)
        has_tensor_weights = isinstance(self.weights, dict) and any(k.endswith(".weight") for k in self.weights)
        if self.weights and (not isinstance(self.weights, dict) or "path" in self.weights or has_tensor_weights):
            self.load_weights()
        else:
            self.init_weights(seed)

    # -- parameters -----------------------------------------------------------------------------
    def init_weights(self, seed: int = None):
        """zero_dce.py:152-158: conv weight ~ N(0, 0.02); bias keeps torch's Conv2d default
        U(+-1/sqrt(fan_in))."""
        g = torch.Generator()
        if seed is not None:
            g.manual_seed(int(seed))
        else:
            g.seed()
        for name in _LAYERS:
            w = self._params[f"{name}.weight"]
            self._params[f"{name}.weight"] = torch.randn(w.shape, generator=g) * 0.02
            bound = 1.0 / (w.shape[1] * 9) ** 0.5
            self._params[f"{name}.bias"] = (torch.rand(w.shape[0], generator=g) * 2 - 1) * bound
        self._on_params_change()

    def _on_params_change(self):
        self._plan_dirty = True

    def _on_device_change(self):
        self._destroy_plan()

    def set_tuning(self, rows_per_block: int = 0, band_rows: int = 0, group_frames: int = 0):
        """Launch-schedule knobs of the plan (0 = automatic); see ie_zdce_set_tuning."""
        self._tuning = (int(rows_per_block), int(band_rows), int(group_frames))
        if self._plan is not None:
            capi.check(capi.lib().ie_zdce_set_tuning(self._plan, *self._tuning), "ie_zdce_set_tuning")

    # -- native plan ----------------------------------------------------------------------------
    def _destroy_plan(self):
        if self._plan is not None:
            capi.lib().ie_zdce_destroy(self._plan)
            self._plan = None
            self._plan_device = None

    def __del__(self):
        try:
            self._destroy_plan()
        except Exception:
            pass

    def _ensure_plan(self):
        if self.device.type != "cuda":
            raise capi.IEError(f"ZeroDCE_RE runs on the HIP device only (model.device = {self.device}); "
                               f"call .to('cuda').  There is no CPU execution path in this engine.")
        L = capi.lib()
        if self._plan is None or self._plan_device != self.device:
            self._destroy_plan()
            cfg = capi.ZdceConfig(self.in_channels, self.num_channels, self.num_iters,
                                  capi.DTYPES[self.compute_dtype], self.device.index or 0, self.variant)
            h = C.c_void_p()
            capi.check(L.ie_zdce_create(C.byref(cfg), C.byref(h)), "ie_zdce_create")
            self._plan, self._plan_device, self._plan_dirty = h, self.device, True
            capi.check(L.ie_zdce_set_tuning(self._plan, *self._tuning), "ie_zdce_set_tuning")
        if self._plan_dirty:
            for k, name in enumerate(_LAYERS, start=1):
                w = self._params[f"{name}.weight"].contiguous()
                b = self._params[f"{name}.bias"].contiguous()
                capi.check(L.ie_zdce_set_layer(self._plan, k, capi.ptr(w), capi.ptr(b), w.shape[0], w.shape[1]),
                           f"ie_zdce_set_layer({k})")
            with torch.cuda.device(self.device):
                capi.check(L.ie_zdce_finalize(self._plan, capi.current_stream()), "ie_zdce_finalize")
            self._plan_dirty = False
        return self._plan

    # -- forward --------------------------------------------------------------------------------
    def _prep(self, x: torch.Tensor) -> torch.Tensor:
        if x.dim() == 3:
            x = x[None]
        if x.dim() != 4 or x.shape[1] != self.in_channels:
            raise ValueError(f"`image` must be (N,{self.in_channels},H,W), but got {tuple(x.shape)}.")
        return x.to(self.device, torch.float32).contiguous()

    def enhance(self, image: torch.Tensor, want_enhanced: bool = True, want_adjust: bool = False,
                want_u8: bool = False, out: dict = None) -> dict:
        """Low-level entry: NCHW fp32 frames in, any of enhanced (N,3,H,W) fp32 / adjust
        (N,24,H,W) fp32 / u8 (N,H,W,3) uint8 out.  ``out`` may carry preallocated tensors."""
        plan = self._ensure_plan()
        x = self._prep(image)
        n, _, h, w = x.shape
        self._last_shape = (n, h, w)
        out = dict(out or {})
        dev = x.device
        if want_enhanced and "enhanced" not in out:
            out["enhanced"] = torch.empty((n, 3, h, w), device=dev, dtype=torch.float32)
        if want_adjust and "adjust" not in out:
            out["adjust"] = torch.empty((n, self.out_channels, h, w), device=dev, dtype=torch.float32)
        if want_u8 and "u8" not in out:
            out["u8"] = torch.empty((n, h, w, 3), device=dev, dtype=torch.uint8)
        with torch.cuda.device(dev):
            capi.check(capi.lib().ie_zdce_forward(plan, capi.ptr(x), capi.ptr(out.get("enhanced")),
                                                  capi.ptr(out.get("adjust")), capi.ptr(out.get("u8")), n, h, w,
                                                  capi.current_stream()), "ie_zdce_forward")
        return out

    def enhance_u8(self, image_u8: torch.Tensor, want_enhanced: bool = False, want_u8: bool = True, out: dict = None) -> dict:
        """uint8 HWC frames in -- (N,H,W,3) or (H,W,3), as read from disk (core/image/io.py:82-89) -- uint8 HWC frames out under the
        save_image rule (and / or the fp32 ``enhanced``): predict.py:76-105's read -> /255 -> forward -> save_image in one call."""
        plan = self._ensure_plan()
        x = image_u8[None] if image_u8.dim() == 3 else image_u8
        if x.dim() != 4 or x.shape[3] != 3 or x.dtype != torch.uint8:
            raise ValueError(f"`image_u8` must be uint8 (N,H,W,3), but got {tuple(x.shape)} {x.dtype}.")
        x = x.to(self.device).contiguous()
        n, h, w, _ = x.shape
        self._last_shape = (n, h, w)
        out = dict(out or {})
        if want_enhanced and "enhanced" not in out:
            out["enhanced"] = torch.empty((n, 3, h, w), device=x.device, dtype=torch.float32)
        if want_u8 and "u8" not in out:
            out["u8"] = torch.empty((n, h, w, 3), device=x.device, dtype=torch.uint8)
        with torch.cuda.device(x.device):
            capi.check(capi.lib().ie_zdce_forward_u8(plan, capi.ptr(x), capi.ptr(out.get("enhanced")), None, capi.ptr(out.get("u8")), n, h, w,
                                                     capi.current_stream()), "ie_zdce_forward_u8")
        return out

    def forward(self, datapoint: dict, *args, **kwargs) -> dict:
        """zero_dce.py:173-191.  Returns ``adjust_0..7`` (views of ``adjust``), ``adjust`` and
        ``enhanced`` -- ``enhanced`` is inserted last (predict.py:97 pops the last item)."""
        self.assert_datapoint(datapoint)
        res = self.enhance(datapoint.get("image"), want_enhanced=True, want_adjust=True)
        outputs = OrderedDict()
        for i, r in enumerate(torch.split(res["adjust"], 3, dim=1)):
            outputs[f"adjust_{i}"] = r
        outputs["adjust"] = res["adjust"]
        outputs["enhanced"] = res["enhanced"]
        return outputs

    @property
    def macs_per_pixel(self) -> int:
        """Multiply-accumulates per pixel of the seven convolutions (79,200 at the default widths)."""
        c, ci, co = self.num_channels, self.in_channels, self.in_channels * self.num_iters
        return 9 * (ci * c + 3 * c * c + 2 * 2 * c * c + 2 * c * co)

    def _require_cuda(self):
        if self.device.type != "cuda":
            raise capi.IEError(f"{self.__class__.__name__} runs on the HIP device only (model.device = {self.device}); call .to('cuda').")

    supports_mixed_training = True

    def forward_train(self, x, store, tape):
        """Training graph (zero_dce.py:173-191) on the taped building blocks with saved activations: the fused inference plan
        keeps x1/x4/x6 in LDS and cannot be differentiated, so training runs the layers one by one -- fp32 kernels (parity
        grade) or, with the trainer's ``compute_dtype`` bf16 / f16, the generic 16-bit matrix-core kernels; returns ``adjust``
        (NHWC), ``enhanced`` (NCHW, last) and the NCHW image for the zero-reference loss."""
        from . import ops, train as T

        def conv(k, srcs, act):
            name = f"e_conv{k}"
            return T.t_conv(tape, srcs, store.conv(name), *store.conv_grads(name), act=act)
        image_nhwc = x
        if store.compute_dtype != "f32":
            # 16-bit graph: the seven layers (forward, data and weight gradients) on the generic matrix-core kernels, RGB carried
            # as 16 channels; the curve maps return to fp32 for the LE curve and the zero-reference loss
            x = T.t_cast(tape, x, ops.TORCH_DT[store.compute_dtype], channels=16)
        x1 = conv(1, [x], "relu")
        x2 = conv(2, [x1], "relu")
        x3 = conv(3, [x2], "relu")
        x4 = conv(4, [x3], "relu")
        x5 = conv(5, [x3, x4], "relu")
        x6 = conv(6, [x2, x5], "relu")
        xr = conv(7, [x1, x6], "tanh")
        if xr.dtype != torch.float32:
            xr = T.t_cast(tape, xr, torch.float32)
        image = ops.to_nchw(image_nhwc)
        enhanced = T.t_curve(tape, image, xr, self.num_iters)
        return {"adjust": xr, "enhanced": enhanced, "_image": image}

    def feature(self, idx: int) -> torch.Tensor:
        """Intermediate x{idx} (1..6) of the last forward as NCHW fp32 (unit parity / debugging)."""
        plan = self._ensure_plan()
        L = capi.lib()
        n, h, w = self._last_shape
        t = torch.empty((n, self.num_channels, h, w), device=self.device, dtype=torch.float32)
        with torch.cuda.device(self.device):
            capi.check(L.ie_zdce_get_feature(plan, idx, capi.ptr(t), capi.current_stream()), "ie_zdce_get_feature")
        return t
