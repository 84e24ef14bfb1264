"""MI355X-native hot path of SKKUAutoLab/ETSS-07-ImageEnhance (``mon``).

Host-side mirror of the reference's model-registry / operator API
(``mon.MODELS.build(...)`` -> model with ``forward(datapoint) -> dict`` and
``infer()``; src/mon/core/factory.py, src/mon/vision/model.py) over a C-ABI HIP
library (``libie_hip.so``, declared in ``include/ie_hip.h``).  PyTorch tensors
are containers only; all arithmetic on the path runs in hand-written gfx950
kernels.  There is no CPU fallback: importing is cheap, but building or running
a model without the built library / a GPU raises.
"""
from .registry import Factory, ModelFactory, MODELS
from .globals_ import Task, Scheme
from . import capi
from .model import Model, VisionModel, ImageEnhancementModel
from .zero_dce import ZeroDCE_RE
from . import ops
from .generators import ZeroDCEpp_RE, LLUnetpp_RE, LLLINet, UNet, ZeroIGEnhance, ZeroIG, RRDNet_RE, ZSN2N, ZeroDCEV, GCENet, RetinexNet, HINet_RE
from . import image
from . import video
from . import sharding
from . import losses, train
from .losses import LOSSES
from .train import Trainer, ModelCheckpoint, get_latest_checkpoint

__version__ = "0.1.0"
__all__ = ["Factory", "ModelFactory", "MODELS", "Task", "Scheme", "capi", "Model", "VisionModel",
           "ImageEnhancementModel", "ZeroDCE_RE", "ZeroDCEpp_RE", "LLUnetpp_RE", "LLLINet", "UNet", "ZeroIGEnhance", "ZeroIG", "RRDNet_RE", "ZSN2N", "ZeroDCEV", "GCENet", "RetinexNet", "HINet_RE",
           "ops", "image", "video", "sharding", "losses", "train", "LOSSES", "Trainer", "ModelCheckpoint", "get_latest_checkpoint"]
