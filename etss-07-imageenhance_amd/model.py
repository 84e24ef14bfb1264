"""Model base classes: host-side mirror of ``mon.nn.Model`` (src/mon/nn/model.py:172-1037),
``VisionModel`` (src/mon/vision/model.py:28-131) and ``ImageEnhancementModel``
(src/mon/vision/enhance/base.py:28-128) -- only the inference/operator surface: construction
from registry kwargs, weight assignment/loading, ``forward``/``infer`` dict contract.

The reference classes are ``LightningModule``s whose arithmetic is ATen; here a model owns
host fp32 parameters plus a native plan in ``libie_hip.so`` and there is no CPU execution path.
"""
from __future__ import annotations

import time
from abc import ABC, abstractmethod
from collections import OrderedDict
from typing import Any, Sequence

import torch

from . import capi
from .globals_ import Scheme, Task
from .registry import kebabize
from .weights import read_state_dict


def _is_image(v) -> bool:
    return isinstance(v, torch.Tensor) and v.dim() in (3, 4) and v.dtype.is_floating_point


def get_image_size(x, divisor: int = None) -> tuple[int, int]:
    """core/image/utils.py:332-384 for tensors / ints / (h, w) sequences; ceil to ``divisor``."""
    import math
    if isinstance(x, torch.Tensor):
        size = (x.shape[-2], x.shape[-1])
    elif isinstance(x, (int, float)):
        size = (int(x), int(x))
    elif isinstance(x, (list, tuple)):
        if len(x) == 3:
            size = tuple(x[0:2]) if x[0] >= x[2] else tuple(x[1:3])
        elif len(x) == 2:
            size = tuple(x)
        elif len(x) == 1:
            size = (x[0], x[0])
        else:
            raise ValueError(f"`input` must be a `list` of length in range ``[1, 3]``, but got {x}.")
    else:
        raise TypeError(f"`input` must be a tensor, an int or a sequence of ints, but got {type(x)}.")
    if divisor is not None:
        size = (int(math.ceil(size[0] / divisor) * divisor), int(math.ceil(size[1] / divisor) * divisor))
    return size


class Model(ABC):
    """nn/model.py:172-277: registry-constructed object with ``weights`` handling.  Unknown
    keyword arguments (``root``, ``fullname``, ``loss``, ``metrics``, ``optimizers``, ``debug``,
    ``verbose`` ...) are accepted and ignored, as the reference swallows them via ``**kwargs``."""

    arch: str = ""
    tasks: list[Task] = []
    schemes: list[Scheme] = []
    zoo: dict = {}

    def __init__(self, name: str = None, fullname: str = None, root: Any = None, in_channels: int = 3,
                 out_channels: int = None, num_classes: int = None, weights: Any = None, loss: Any = None,
                 metrics: Any = None, optimizers: Any = None, debug: bool = True, verbose: bool = True,
                 device: Any = None, *args, **kwargs):
        self.debug, self.verbose = debug, verbose
        self.name = name
        self.fullname = fullname or self.name
        self.root = root
        self.in_channels = in_channels
        self.out_channels = out_channels or num_classes or self.in_channels
        self.weights = None
        self.assign_weights(weights)
        self.training = True
        self.predicting = False
        self._device = torch.device(device) if device is not None else torch.device("cpu")
        self._params: "OrderedDict[str, torch.Tensor]" = OrderedDict()

    # -- properties ---------------------------------------------------------------------------
    @property
    def name(self) -> str:
        return self._name

    @name.setter
    def name(self, name: str):
        if name is None or name == "":
            name = kebabize(self.__class__.__name__).lower()  # nn/model.py:286-290
        self._name = name

    @property
    def device(self) -> torch.device:
        return self._device

    @property
    def params(self) -> int:
        return sum(int(v.numel()) for v in self._params.values())

    # -- torch.nn.Module-like surface used by predict.py:46-48 -------------------------------
    def to(self, device):
        dev = torch.device(device)
        if dev.type == "cuda" and dev.index is None:
            dev = torch.device("cuda", torch.cuda.current_device())
        if dev != self._device:
            self._device = dev
            self._on_device_change()
        return self

    def cuda(self, device=None):
        return self.to("cuda" if device is None else f"cuda:{int(device)}")

    def eval(self):
        self.training = False
        return self

    def train(self, mode: bool = True):
        self.training = bool(mode)
        return self

    def __call__(self, *args, **kwargs):
        return self.forward(*args, **kwargs)

    def _on_device_change(self):
        pass

    # -- weights (nn/model.py:381-426) -------------------------------------------------------
    def assign_weights(self, weights: Any, overwrite: bool = False):
        if weights is None:
            pass
        elif isinstance(weights, str) and weights in self.zoo:
            weights = self.zoo[weights]  # zoo entries need a local "path"; downloads are unavailable offline
        elif isinstance(weights, (str, bytes)) or hasattr(weights, "__fspath__"):
            from .weights import is_weights_file
            if not is_weights_file(weights):
                raise ValueError(f"`weights` must be a valid path to a weight file, but got {weights}.")
        elif isinstance(weights, dict):
            pass
        else:
            raise TypeError(f"`weights` must be a zoo key, a path or a dict, but got {type(weights)}.")
        self.weights = weights or self.weights

    def load_weights(self, weights: Any = None, overwrite: bool = False):
        self.assign_weights(weights, overwrite)
        state_dict = read_state_dict(self.weights) if self.weights else None
        if state_dict:
            self.load_state_dict(state_dict)

    def expected_keys(self) -> "OrderedDict[str, tuple]":
        """name -> shape of every parameter this model consumes."""
        return OrderedDict((k, tuple(v.shape)) for k, v in self._params.items())

    def load_state_dict(self, state_dict: dict, strict: bool = True):
        """Strict on the model's own keys like ``nn.Module.load_state_dict`` (nn/model.py:424);
        ``loss.*`` entries that mon checkpoints carry are ignored."""
        exp = self.expected_keys()
        sd = {self.map_key(k): v for k, v in state_dict.items()}
        missing = [k for k in exp if k not in sd]
        unexpected = [k for k in sd if k not in exp and not k.startswith("loss.")]
        bad_shape = [f"{k}: {tuple(sd[k].shape)} vs {exp[k]}" for k in exp if k in sd and tuple(sd[k].shape) != exp[k]]
        if bad_shape or (strict and (missing or unexpected)):
            msgs = []
            if missing: msgs.append(f"Missing key(s) in state_dict: {missing}")
            if unexpected: msgs.append(f"Unexpected key(s) in state_dict: {unexpected}")
            if bad_shape: msgs.append(f"size mismatch: {bad_shape}")
            raise RuntimeError(f"Error(s) in loading state_dict for {self.__class__.__name__}: " + "; ".join(msgs))
        for k in exp:
            if k in sd:
                self._params[k] = torch.as_tensor(sd[k]).detach().to("cpu", torch.float32).contiguous().clone()
        self._on_params_change()

    def map_key(self, k: str) -> str:
        return k

    def state_dict(self) -> "OrderedDict[str, torch.Tensor]":
        return OrderedDict((k, v.clone()) for k, v in self._params.items())

    def _on_params_change(self):
        pass

    @abstractmethod
    def init_weights(self, seed: int = None):
        ...

    @abstractmethod
    def forward(self, datapoint: dict, *args, **kwargs) -> dict:
        ...


class VisionModel(Model, ABC):
    """vision/model.py:28-131."""

    def assert_datapoint(self, datapoint: dict):
        pass

    def assert_outputs(self, outputs: dict):
        pass

    def compute_efficiency_score(self, image_size: int | Sequence[int] = 512, channels: int = 3, runs: int = 100, verbose: bool = False):
        """vision/model.py:36-70: (FLOPs, params, average seconds per forward) on one random ``image_size`` frame.  FLOPs follow
        the reference's thop convention (one multiply-accumulate of a convolution = 1): counted by the conv wrappers during one
        forward (``macs_per_pixel`` for models whose forward is one fused plan); time = ``runs`` forwards bracketed by device syncs."""
        from . import ops
        h, w = get_image_size(image_size)
        datapoint = {"image": torch.rand(1, channels, h, w).to(self.device)}
        ops._mac_counter = [0]
        try:
            self.forward(dict(datapoint))
            flops = float(ops._mac_counter[0])
        finally:
            ops._mac_counter = None
        if flops == 0 and hasattr(self, "macs_per_pixel"):
            flops = float(self.macs_per_pixel) * h * w
        if self.device.type == "cuda":
            torch.cuda.synchronize(self.device)
        t0 = time.perf_counter()
        for _ in range(runs):
            self.forward(dict(datapoint))
        if self.device.type == "cuda":
            torch.cuda.synchronize(self.device)
        avg_time = (time.perf_counter() - t0) / max(1, runs)
        if verbose:
            print(f"FLOPs (G) : {flops / 1e9:.4f}\nParams (M): {self.params / 1e6:.4f}\nTime (s)  : {avg_time:.17f}")
        return flops, self.params, avg_time

    def capture_graph(self, image: torch.Tensor, warmup: int = 3):
        """Serving path: capture ``forward`` for one input shape into a HIP graph (one ``hipGraphLaunch`` per call instead of
        the ~50-150 kernel launches a generator enqueues from Python; the launch-bound case is small batches).  Returns
        ``run(image) -> outputs``: the input is copied into the captured buffer, the graph replayed, and the SAME output
        tensors returned on every call (consume or clone them before the next call).  Weights are baked in: capture again
        after ``load_state_dict``.  The reference has no counterpart (it relies on eager PyTorch)."""
        if self.device.type != "cuda":
            raise capi.IEError("capture_graph needs the model on the HIP device")
        static_in = image.to(self.device, torch.float32).contiguous().clone()
        stream = torch.cuda.Stream(device=self.device)
        stream.wait_stream(torch.cuda.current_stream(self.device))
        with torch.cuda.stream(stream):                  # warm-up on a side stream: weight packing, plan workspaces, attributes
            for _ in range(max(1, warmup)):
                self.forward({"image": static_in})
        torch.cuda.current_stream(self.device).wait_stream(stream)
        torch.cuda.synchronize(self.device)
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph):
            outputs = self.forward({"image": static_in})

        def run(x: torch.Tensor) -> dict:
            if tuple(x.shape) != tuple(static_in.shape):
                raise ValueError(f"the graph was captured for {tuple(static_in.shape)}, but got {tuple(x.shape)}.")
            static_in.copy_(x, non_blocking=True)
            graph.replay()
            return outputs
        run.graph, run.static_input = graph, static_in
        return run

    def _resize(self, v: torch.Tensor, size: tuple[int, int]) -> torch.Tensor:
        """core.resize on tensors == F.interpolate(bilinear, align_corners=False)
        (core/image/geometry/transform.py:216-223), run by the HIP kernel."""
        squeeze = v.dim() == 3
        x = v[None] if squeeze else v
        x = x.to(self.device, torch.float32).contiguous()
        n, c, h, w = x.shape
        if (h, w) == tuple(size):
            return v
        out = torch.empty((n, c, size[0], size[1]), device=x.device, dtype=torch.float32)
        with torch.cuda.device(x.device):          # the current-stream lookup must be this device's
            capi.check(capi.lib().ie_resize_bilinear_nchw_f32(capi.ptr(x), capi.ptr(out), n, c, h, w, size[0], size[1], 0,
                                                              capi.current_stream()), "ie_resize_bilinear_nchw_f32")
        return out[0] if squeeze else out

    def infer(self, datapoint: dict, image_size: int | Sequence[int] = 512, resize: bool = False, *args, **kwargs) -> dict:
        """vision/model.py:76-127: resize every image entry to ``image_size`` (``resize=True``) or up
        to the next multiple of 32, move tensors to the device, time ``forward``, resize image
        outputs back, add ``"time"``."""
        self.assert_datapoint(datapoint)
        image = datapoint.get("image")
        h0, w0 = get_image_size(image)
        for k, v in datapoint.items():
            if isinstance(v, torch.Tensor):
                datapoint[k] = v.to(self.device)
        for k, v in datapoint.items():
            if _is_image(v):
                datapoint[k] = self._resize(v, get_image_size(image_size) if resize else get_image_size(v, 32))
        if self.device.type == "cuda":
            torch.cuda.synchronize(self.device)
        t0 = time.perf_counter()
        outputs = self.forward(datapoint, *args, **kwargs)
        if self.device.type == "cuda":
            torch.cuda.synchronize(self.device)   # the reference does not sync; we time the kernels
        dt = time.perf_counter() - t0
        self.assert_outputs(outputs)
        for k, v in list(outputs.items()):
            if _is_image(v):
                h1, w1 = get_image_size(v)
                if h1 != h0 or w1 != w0:
                    outputs[k] = self._resize(v, (h0, w0))
        outputs["time"] = dt
        return outputs


class ImageEnhancementModel(VisionModel, ABC):
    """vision/enhance/base.py:28-57."""

    def assert_datapoint(self, datapoint: dict):
        if "image" not in datapoint:
            raise ValueError("The key ``'image'`` must be defined in the `datapoint`.")
        has_target = any(s in self.schemes for s in [Scheme.SUPERVISED]) and not self.predicting
        if has_target and "ref_image" not in datapoint:
            raise ValueError("The key ``'ref_image'`` must be defined in the `datapoint`.")

    def assert_outputs(self, outputs: dict):
        if "enhanced" not in outputs:
            raise ValueError("The key ``'enhanced'`` must be defined in the `outputs`.")
