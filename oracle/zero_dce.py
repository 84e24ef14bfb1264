"""Oracle (TEST INFRASTRUCTURE): CPU fp32 restatement of the Zero-DCE forward.

Follows
  * ``src/mon/vision/enhance/llie/zero_dce/zero_dce.py:132-139`` (layers) and
    ``:173-191`` (forward: 7 conv3x3 + ReLU, channel concats [x3,x4], [x2,x5],
    [x1,x6], tanh, 8-step LE curve ``y = y + r*(y^2 - y)``), and
  * ``src/mon_extra/vision/enhance/llie/zero_dce/model.py:28-50`` (the original
    network the mon class re-implements; same layer names ``e_conv1..7``).

Pinned by ``tests/golden/zero_dce_*.npz`` (outputs of the mon_extra reference
file run in the build container, generator: ``tests/golden/make_golden.py``).

The uint8 rule is torchvision ``save_image`` as called from
``src/mon/core/image/io.py:139-140``: ``clamp(x*255 + 0.5, 0, 255) -> uint8``.
torchvision is absent here, so that rule is restated from its documented
behaviour ("parity unpinned" for the uint8 edge only).
"""
from __future__ import annotations

import torch
import torch.nn.functional as F

LAYERS = ("e_conv1", "e_conv2", "e_conv3", "e_conv4", "e_conv5", "e_conv6", "e_conv7")


def layer_shapes(in_channels: int = 3, num_channels: int = 32, num_iters: int = 8):
    """(Cout, Cin) per layer; zero_dce.py:133-139."""
    c, o = num_channels, in_channels * num_iters
    return [(c, in_channels), (c, c), (c, c), (c, c), (c, 2 * c), (c, 2 * c), (o, 2 * c)]


def init_state_dict(seed: int = 1, in_channels: int = 3, num_channels: int = 32,
                    num_iters: int = 8, w_std: float = 0.02, gain: float = 1.0) -> dict:
    """Seeded weights under the reference init rule (zero_dce.py:152-158):
    conv weight ~ N(0, 0.02); bias keeps torch's default U(+-1/sqrt(fan_in)).
    ``gain`` > 1 gives a "trained-like" variant whose tanh output is not tiny."""
    g = torch.Generator().manual_seed(seed)
    sd = {}
    for name, (co, ci) in zip(LAYERS, layer_shapes(in_channels, num_channels, num_iters)):
        sd[f"{name}.weight"] = torch.randn(co, ci, 3, 3, generator=g) * (w_std * gain)
        bound = 1.0 / (ci * 9) ** 0.5
        sd[f"{name}.bias"] = (torch.rand(co, generator=g) * 2 - 1) * bound
    return sd


def forward(sd: dict, x: torch.Tensor, num_iters: int = 8, return_features: bool = False) -> dict:
    """x: (N,3,H,W) fp32 in [0,1].  Returns the reference's output dict
    (zero_dce.py:183-191): adjust_i, adjust (N,3*iters,H,W), enhanced (last)."""
    def conv(name, t):
        return F.conv2d(t, sd[f"{name}.weight"], sd[f"{name}.bias"], stride=1, padding=1)
    x1 = F.relu(conv("e_conv1", x))
    x2 = F.relu(conv("e_conv2", x1))
    x3 = F.relu(conv("e_conv3", x2))
    x4 = F.relu(conv("e_conv4", x3))
    x5 = F.relu(conv("e_conv5", torch.cat([x3, x4], 1)))
    x6 = F.relu(conv("e_conv6", torch.cat([x2, x5], 1)))
    x_r = torch.tanh(conv("e_conv7", torch.cat([x1, x6], 1)))
    x_rs = torch.split(x_r, 3, dim=1)
    y = x
    out = {}
    for i in range(num_iters):
        y = y + x_rs[i] * (torch.pow(y, 2) - y)
        out[f"adjust_{i}"] = x_rs[i]
    out["adjust"] = x_r
    out["enhanced"] = y
    if return_features:
        out["_features"] = (x1, x2, x3, x4, x5, x6)
    return out


def curve(x: torch.Tensor, x_r: torch.Tensor, num_iters: int = 8) -> torch.Tensor:
    """LE curve alone (zero_dce.py:186-188)."""
    y = x
    for r in torch.split(x_r, 3, dim=1)[:num_iters]:
        y = y + r * (torch.pow(y, 2) - y)
    return y


def to_uint8_save_image(x: torch.Tensor) -> torch.Tensor:
    """torchvision.utils.save_image quantisation (core/image/io.py:139-140):
    ``x.mul(255).add_(0.5).clamp_(0, 255).to(uint8)``; NCHW float -> NHWC u8."""
    return x.mul(255).add(0.5).clamp(0, 255).to(torch.uint8).permute(0, 2, 3, 1).contiguous()


def to_uint8_truncate(x: torch.Tensor) -> torch.Tensor:
    """Debug/video path ``(x*255).astype(uint8)`` (core/image/utils.py:795):
    C-style truncation of an unclamped float.  Out-of-range values are
    implementation defined in numpy; the oracle clamps first (documented)."""
    return x.mul(255).clamp(0, 255).to(torch.uint8).permute(0, 2, 3, 1).contiguous()


FLOP_PER_PX = 2 * (27 * 32 + 3 * 288 * 32 + 2 * 576 * 32 + 576 * 24)  # 158,400
