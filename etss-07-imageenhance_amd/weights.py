"""Checkpoint reading: mirror of ``load_state_dict`` (src/mon/nn/model.py:99-131) and
``Path.is_weights_file`` (src/mon/core/pathlib.py:215-223)."""
from __future__ import annotations

import os
from typing import Any

from .globals_ import WEIGHTS_FILE_FORMATS


def is_weights_file(path: Any, exist: bool = True) -> bool:
    p = str(path)
    return (os.path.isfile(p) if exist else True) and os.path.splitext(p)[1].lower() in WEIGHTS_FILE_FORMATS


def read_state_dict(weights: Any) -> dict | None:
    """``weights``: a state_dict, ``{"path": ...}``, or a ``.pt/.pth`` path.  A ``.ckpt``/``.tar``
    path is rejected like the reference does (nn/model.py:401-405).  A Lightning checkpoint's
    ``{"state_dict": ...}`` nesting is unwrapped (nn/model.py:129-130)."""
    import torch
    state_dict = None
    path = None
    if isinstance(weights, dict) and "path" in weights:
        path = weights["path"]
    elif isinstance(weights, dict):
        state_dict = weights
    elif isinstance(weights, (str, os.PathLike)):
        if not is_weights_file(weights):
            raise ValueError(f"`weights` must be a valid path to a weight file, but got {weights}.")
        path = weights
    elif weights is not None:
        raise TypeError(f"`weights` must be a dict or a path, but got {type(weights)}.")
    if path is not None:
        state_dict = torch.load(str(path), map_location="cpu", weights_only=True)
    if state_dict is not None and "state_dict" in state_dict:
        state_dict = state_dict["state_dict"]
    return state_dict
