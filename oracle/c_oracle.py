"""Oracle (TEST INFRASTRUCTURE): ctypes binding of oracle/c/ie_oracle.c.

Double-accumulating plain-C restatement of conv2d / LE-curve / Zero-DCE forward,
used to anchor the torch restatement independently of oneDNN's summation order.
"""
from __future__ import annotations

import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "_build", "libie_oracle.so")
_lib = None

ACT = {"none": 0, "relu": 1, "lrelu": 2, "tanh": 3, "sigmoid": 4}
PAD = {"zeros": 0, "replicate": 1}


def build(force: bool = False) -> str:
    if force or not os.path.exists(_LIB_PATH):
        subprocess.check_call(["make", "-C", _HERE, "-s"] + (["-B"] if force else []))
    return _LIB_PATH


def lib():
    global _lib
    if _lib is None:
        build()
        _lib = ctypes.CDLL(_LIB_PATH)
    return _lib


def _fp(a):
    return a.ctypes.data_as(ctypes.POINTER(ctypes.c_float))


def conv2d(x, w, b=None, stride=1, pad=1, pad_mode="zeros", groups=1, act="none", slope=0.0):
    x = np.ascontiguousarray(x, np.float32)
    w = np.ascontiguousarray(w, np.float32)
    N, C, H, W = x.shape
    Co, _, kh, kw = w.shape
    Ho = (H + 2 * pad - kh) // stride + 1
    Wo = (W + 2 * pad - kw) // stride + 1
    y = np.empty((N, Co, Ho, Wo), np.float32)
    bb = None if b is None else np.ascontiguousarray(b, np.float32)
    rc = lib().ieo_conv2d_nchw_f32(_fp(x), N, C, H, W, _fp(w), None if bb is None else _fp(bb),
                                   Co, kh, kw, stride, pad, PAD[pad_mode], groups, ACT[act],
                                   ctypes.c_float(slope), _fp(y))
    if rc != 0:
        raise ValueError(f"ieo_conv2d_nchw_f32 rc={rc}")
    return y


def zero_dce_forward(sd: dict, x, num_channels=32, num_iters=8):
    x = np.ascontiguousarray(x, np.float32)
    N, _, H, W = x.shape
    ws = [np.ascontiguousarray(np.asarray(sd[f"e_conv{k}.weight"]), np.float32) for k in range(1, 8)]
    bs = [np.ascontiguousarray(np.asarray(sd[f"e_conv{k}.bias"]), np.float32) for k in range(1, 8)]
    PF = ctypes.POINTER(ctypes.c_float)
    wa = (PF * 7)(*[_fp(a) for a in ws])
    ba = (PF * 7)(*[_fp(a) for a in bs])
    adjust = np.empty((N, 3 * num_iters, H, W), np.float32)
    enh = np.empty((N, 3, H, W), np.float32)
    rc = lib().ieo_zero_dce_forward_f32(_fp(x), N, H, W, num_channels, num_iters, wa, ba,
                                        _fp(adjust), _fp(enh))
    if rc != 0:
        raise RuntimeError(f"ieo_zero_dce_forward_f32 rc={rc}")
    return {"adjust": adjust, "enhanced": enh}


def to_u8_round(x):
    x = np.ascontiguousarray(x, np.float32)
    N, C, H, W = x.shape
    o = np.empty((N, H, W, C), np.uint8)
    lib().ieo_to_u8_round(_fp(x), N, C, H, W, o.ctypes.data_as(ctypes.POINTER(ctypes.c_uint8)))
    return o
