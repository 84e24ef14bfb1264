"""Oracle (TEST INFRASTRUCTURE): CPU fp32 restatements (plain torch.nn.functional) of the other generators.

  zero_dcepp_forward   src/mon/vision/enhance/llie/zero_dce/zero_dcepp.py:173-208 (DSConv2d dsconv.py:122-149)
                       PINNED by tests/golden/zero_dcepp_*.npz (reference file mon_extra/.../zero_dce++/model.py)
  llunetpp_forward     src/mon/vision/enhance/llie/llunetpp/llunetpp.py:97-137, 218-243
                       PINNED by tests/golden/llunetpp_*.npz (reference file mon_extra/.../llunet++/model.py)
  lllinet_forward      src/mon/vision/enhance/llie/lllinet/lllinet.py:97-260 (+ normalization.py:177-224, attention.py:1156-1182): the BLOCK is PINNED (round 4) by
                       tests/golden/lllinet_w2_*.npz (UNetConvBlock, LearnableInstanceNorm2d and SimAM executed out of their files; grid order transcribed, widths 2 .. 32)
                       and its norm leaf by tests/golden/learnable_instance_norm.npz.
  lllinet_hvi_forward  lllinet.py:262-388 + core/image/color.py:313-434 (RGBToHVI).  rgb_to_hvi / hvi_to_rgb are PINNED (round 4) by tests/golden/hvi_transform_*.npz: the class
                       RGBToHVI executed out of color.py's syntax tree (the module imports cv2 and kornia, the class is pure torch); the LLLINet grid around it: parity unpinned.
  unet_forward         src/mon/vision/segment/unet/unet.py:35-197 with line 51's typo read as conv1; eval-mode BN.
                       PINNED by tests/golden/unet_mtfe_*.npz: the reference tree's pure-torch twin of the same topology,
                       mon_extra/vision/enhance/llie/mtfe/unet_model.py + unet_parts.py (bilinear and ConvTranspose2d decoders,
                       convs with bias, keys ``*.double_conv.{0,1,3,4}`` -> unet_from_mtfe_keys), run in the build container.
  unet_ps variant      + Upsampler conv3x3 + nn.PixelShuffle(2) (nn/modules/encoder/edsr.py:88-123): the encoder / DoubleConv /
                       pad / cat / head around it are the pinned ones; the upsampler leaf is ``upsampler_forward`` below, PINNED by
                       tests/golden/upsampler_edsr.npz (round 4: edsr.py imports torch only and loads as a module).
  zero_ig_enhance      src/mon/vision/enhance/llie/zero_ig/zero_ig.py:307-348 (Enhance)
  zero_ig_denoise      zero_ig.py:257-304 (Denoise1 / Denoise2)
  zero_ig_forward      zero_ig.py:434-452 (ZeroIG.forward, predicting branch)
                       all three PINNED by tests/golden/zero_ig_*.npz (reference file mon_extra/.../zero_ig/model.py:
                       Enhancer, Denoise_1, Denoise_2 and Finetunemodel.forward run in the build container)
  mprnet_forward       src/mon/vision/enhance/multitask/mprnet/mprnet.py:34-357, 483-553: blocks PINNED (round 4) by tests/golden/mprnet_c8_*.npz (the
                       block classes executed out of mprnet.py); the call order of forward() transcribed.
  hinet_forward        src/mon/vision/enhance/multitask/hinet/hinet.py:34-198, 322-362
                       PINNED by tests/golden/hinet_*.npz (reference file mon_extra/.../hinet/basicsr/models/archs/hinet_arch.py)
  rrdnet_forward       src/mon/vision/enhance/llie/rrdnet/rrdnet.py:208-289: the three stacks + sigmoid / sigmoid / tanh are PINNED by
                       tests/golden/rrdnet_*.npz (reference file mon_extra/vision/enhance/llie/rrdnet/model/RRDNet.py, same
                       Sequential keys); the recombination of :271-279 is five torch ops on those pinned maps.
  zsn2n_forward        src/mon/vision/enhance/denoise/zsn2n/zsn2n.py:73-124: parity unpinned (same).
  retinexnet_forward   src/mon_extra/vision/enhance/llie/retinexnet/model.py:21-90,199-238; single Decom pass, S = R * I_delta.
  retinex_decom / retinexnet_losses / retinex_smooth: the same file's DecomNet, RetinexNet.forward (both losses, nine terms), smooth.
                       PINNED (round 4) by tests/golden/retinexnet_c16_*.npz: the file cannot be imported as a module (cv2, thop, mon), so
                       tests/golden/make_golden.py takes the classes DecomNet / RelightNet / RetinexNet out of its syntax tree by name and executes
                       them unmodified (`.cuda()` the identity) -- RetinexNet.forward(input_low, input_high) as shipped, at channel = 16.
"""
from __future__ import annotations

import torch
import torch.nn.functional as F


def _c(sd, name, x, stride=1, padding=1, groups=1, mode="zeros"):
    w, b = sd[f"{name}.weight"], sd.get(f"{name}.bias")
    if mode == "replicate" and padding > 0:
        x = F.pad(x, (padding,) * 4, mode="replicate")
        padding = 0
    return F.conv2d(x, w, b, stride=stride, padding=padding, groups=groups)


# ---- PixelShuffle up-sampler (the config-3 decoder variant's leaf) ------------------------------------------------
def upsampler_forward(sd: dict, x: torch.Tensor, relu: bool = False) -> torch.Tensor:
    """Upsampler (src/mon/nn/modules/encoder/edsr.py:88-123) for scale = 2^n: per stage conv3x3(n_feats -> 4 n_feats, padding 1) -> nn.PixelShuffle(2) [-> ReLU];
    ``sd`` holds the nn.Sequential's own keys ("0.weight", "0.bias", then "2." or "3." ... for the next stage).  PINNED by tests/golden/upsampler_edsr.npz
    (the file imports torch only: loaded as a module by make_golden.py)."""
    for k in sorted({k.rsplit(".", 1)[0] for k in sd}, key=lambda t: int(t)):
        x = F.pixel_shuffle(_c(sd, k, x), 2)
        if relu:
            x = F.relu(x)
    return x


# ---- Zero-DCE++ -----------------------------------------------------------------------------------------------
def zero_dcepp_forward(sd: dict, x: torch.Tensor, num_iters: int = 8, scale_factor: float = 1.0) -> dict:
    def ds(k, t):
        t = _c(sd, f"e_conv{k}.dw_conv", t, groups=t.shape[1])
        return _c(sd, f"e_conv{k}.pw_conv", t, padding=0)
    xd = x if scale_factor == 1 else F.interpolate(x, scale_factor=1 / scale_factor, mode="bilinear")
    x1 = F.relu(ds(1, xd)); x2 = F.relu(ds(2, x1)); x3 = F.relu(ds(3, x2)); x4 = F.relu(ds(4, x3))
    x5 = F.relu(ds(5, torch.cat([x3, x4], 1)))
    x6 = F.relu(ds(6, torch.cat([x2, x5], 1)))
    xr = torch.tanh(ds(7, torch.cat([x1, x6], 1)))
    if scale_factor != 1:
        xr = F.interpolate(xr, scale_factor=scale_factor, mode="bilinear", align_corners=True)   # nn.UpsamplingBilinear2d
    y = x
    for _ in range(num_iters):
        y = y + xr * (torch.pow(y, 2) - y)
    return {"adjust": xr, "enhanced": y}


# ---- LLUnet++ -------------------------------------------------------------------------------------------------
def _llunet_block(sd, b, x, qa=None, qw=None):
    qa = qa or (lambda t: t)
    qw = qw or (lambda w: w)
    cw = lambda name, t, **kw: F.conv2d(t, qw(sd[f"{name}.weight"]), sd.get(f"{name}.bias"), **kw)
    x12 = qa(cw(f"{b}.conv1_2", x))
    x1 = qa(cw(f"{b}.conv1", x, padding=1))
    x1 = F.instance_norm(x1, weight=sd[f"{b}.norm1.weight"], bias=sd[f"{b}.norm1.bias"], eps=1e-5)
    x1 = qa(F.leaky_relu(x1, 0.2))
    x2 = torch.cat([x1, x12], 1)
    x23 = qa(cw(f"{b}.conv2_3", x2))
    x2 = qa(F.leaky_relu(cw(f"{b}.conv2", x2, padding=1), 0.2))
    x3 = qa(F.leaky_relu(cw(f"{b}.conv3", x2, padding=1), 0.2))
    return qa(x3 + x23)


def llunetpp_forward(sd: dict, x: torch.Tensor, q=None) -> dict:
    """``q = oracle.train.storage_rounding(dtype)``: the engine's 16-bit training graph restated -- every tensor between two kernels
    (conv + fused activation output, norm output, residual sum, resized map) and its gradient rounded to ``dtype``, weights read in
    ``dtype`` -- for tests that separate kernel error from storage rounding.  Default: the plain fp32 / fp64 forward."""
    qa, qw = q if q is not None else (None, None)
    B = lambda n, t: _llunet_block(sd, n, t, qa, qw)
    pool = lambda t: F.max_pool2d(t, 2, 2)
    up = (lambda t: qa(F.interpolate(t, scale_factor=2, mode="bilinear", align_corners=True))) if qa else \
        (lambda t: F.interpolate(t, scale_factor=2, mode="bilinear", align_corners=True))
    cat = lambda *ts: torch.cat(ts, 1)
    if qa:
        x = qa(x.detach().clone())
    x0_0 = B("conv0_0", x)
    x1_0 = B("conv1_0", pool(x0_0))
    x0_1 = B("conv0_1", cat(x0_0, up(x1_0)))
    x2_0 = B("conv2_0", pool(x1_0))
    x1_1 = B("conv1_1", cat(x1_0, up(x2_0)))
    x0_2 = B("conv0_2", cat(x0_0, x0_1, up(x1_1)))
    x3_0 = B("conv3_0", pool(x2_0))
    x2_1 = B("conv2_1", cat(x2_0, up(x3_0)))
    x1_2 = B("conv1_2", cat(x1_0, x1_1, up(x2_1)))
    x0_3 = B("conv0_3", cat(x0_0, x0_1, x0_2, up(x1_2)))
    x4_0 = B("conv4_0", pool(x3_0))
    x3_1 = B("conv3_1", cat(x3_0, up(x4_0)))
    x2_2 = B("conv2_2", cat(x2_0, x2_1, up(x3_1)))
    x1_3 = B("conv1_3", cat(x1_0, x1_1, x1_2, up(x2_2)))
    x0_4 = B("conv0_4", cat(x0_0, x0_1, x0_2, x0_3, up(x1_3)))
    y = F.conv2d(x0_4, (qw or (lambda w: w))(sd["final.weight"]), sd.get("final.bias"))
    if qa:
        y = qa(y)
    return {"enhanced": torch.clamp(y, 0, 1)}


MON_EXTRA_LLUNET_KEYS = ((".conv1_2.", ".conv_1_1_1."), (".conv2_3.", ".conv_1_1."), (".conv1.", ".conv_1."),
                         (".conv2.", ".conv_2."), (".conv3.", ".conv_3."))


def llunetpp_to_extra_keys(sd: dict) -> dict:
    """mon key names -> the original repository's (llunet++/model.py:9-20)."""
    out = {}
    for k, v in sd.items():
        for a, b in MON_EXTRA_LLUNET_KEYS:
            k = k.replace(a, b)
        out[k] = v
    return out


# ---- LLLINet (block pinned by tests/golden/lllinet_w2_*.npz; the model class itself does not import here) ---------------------------
def learnable_instance_norm(x, weight, bias, r, eps=1e-5):
    """src/mon/nn/modules/normalization.py:208-224 (pinned: tests/golden/learnable_instance_norm.npz)."""
    xn = F.instance_norm(x, weight=weight, bias=bias, eps=eps)
    rr = r.reshape(1, -1, 1, 1)
    return xn * rr + x * (1 - rr)


def simam(x, e_lambda=1e-4):
    """src/mon/nn/modules/attention.py:1169-1182."""
    n = x.shape[2] * x.shape[3] - 1
    d = (x - x.mean(dim=[2, 3], keepdim=True)).pow(2)
    v = d.sum(dim=[2, 3], keepdim=True) / n
    return x * torch.sigmoid(d / (4 * (v + e_lambda)) + 0.5)


def _lllinet_block(sd, b, x):
    """src/mon/vision/enhance/llie/lllinet/lllinet.py:133-156."""
    x1 = _c(sd, f"{b}.conv1", x)
    if f"{b}.norm1.r" in sd:
        x1 = learnable_instance_norm(x1, sd[f"{b}.norm1.weight"], sd[f"{b}.norm1.bias"], sd[f"{b}.norm1.r"])
    x1 = simam(F.leaky_relu(x1, 0.2))
    x2 = _c(sd, f"{b}.conv2", x1)
    x3 = torch.cat([x2, _c(sd, f"{b}.conv1_3", x, padding=0)], 1)
    x34 = _c(sd, f"{b}.conv3_4", x3, padding=0)
    x3 = F.leaky_relu(_c(sd, f"{b}.conv3", x3), 0.2)
    return F.leaky_relu(_c(sd, f"{b}.conv4", x3), 0.2) + x34


def lllinet_forward(sd: dict, x: torch.Tensor, clamp: bool = True) -> dict:
    """lllinet.py:229-260 (``clamp=False``: the body alone, for LLLINetHVI which clamps behind its colour transform)."""
    B = lambda n, t: _lllinet_block(sd, n, t)
    pool = lambda t: F.max_pool2d(t, 2, 2)
    up = lambda t: F.interpolate(t, scale_factor=2, mode="bilinear", align_corners=True)
    cat = lambda *ts: torch.cat(ts, 1)
    x0_0 = B("conv0_0", x)
    x1_0 = B("conv1_0", pool(x0_0)); x2_0 = B("conv2_0", pool(x1_0)); x3_0 = B("conv3_0", pool(x2_0)); x4_0 = B("conv4_0", pool(x3_0))
    x3_1 = B("conv3_1", cat(x3_0, up(x4_0)))
    x2_1 = B("conv2_1", cat(x2_0, up(x3_0), up(x3_1)))
    x1_1 = B("conv1_1", cat(x1_0, up(x2_0), up(x2_1)))
    x0_1 = B("conv0_1", cat(x0_0, up(x1_0), up(x1_1)))
    x2_2 = B("conv2_2", cat(x2_0, x2_1, up(x3_1)))
    x1_2 = B("conv1_2", cat(x1_0, x1_1, up(x2_1), up(x2_2)))
    x0_2 = B("conv0_2", cat(x0_0, x0_1, up(x1_1), up(x1_2)))
    x1_3 = B("conv1_3", cat(x1_0, x1_1, x1_2, up(x2_2)))
    x0_3 = B("conv0_3", cat(x0_0, x0_1, x0_2, up(x1_2), up(x1_3)))
    x0_4 = B("conv0_4", cat(x0_0, x0_1, x0_2, x0_3, up(x1_3)))
    y = _c(sd, "final", x0_4, padding=0)
    return {"enhanced": torch.clamp(y, 0, 1) if clamp else y}


# ---- HVI colour transform (core/image/color.py:313-434) and LLLINetHVI (lllinet.py:262-388) ---------------------------------------------
def rgb_to_hvi(image: torch.Tensor, density_k: float = 0.2, eps: float = 1e-8) -> torch.Tensor:
    """color.py:332-361, statement by statement (the masked assignments in the reference's order: the last match wins)."""
    pi = 3.141592653589793
    value = image.max(1)[0]
    img_min = image.min(1)[0]
    hue = torch.zeros_like(value)
    m = image[:, 2] == value
    hue[m] = 4.0 + ((image[:, 0] - image[:, 1]) / (value - img_min + eps))[m]
    m = image[:, 1] == value
    hue[m] = 2.0 + ((image[:, 2] - image[:, 0]) / (value - img_min + eps))[m]
    m = image[:, 0] == value
    hue[m] = (0.0 + ((image[:, 1] - image[:, 2]) / (value - img_min + eps))[m]) % 6
    hue[img_min == value] = 0.0
    hue = hue / 6.0
    saturation = (value - img_min) / (value + eps)
    saturation[value == 0] = 0
    hue, saturation, value = hue.unsqueeze(1), saturation.unsqueeze(1), value.unsqueeze(1)
    k = torch.full([1], density_k)
    color_sensitive = ((value * 0.5 * pi).sin() + eps).pow(k)
    cx, cy = (2.0 * pi * hue).cos(), (2.0 * pi * hue).sin()
    return torch.cat([color_sensitive * saturation * cx, color_sensitive * saturation * cy, value], dim=1)


def hvi_to_rgb(image: torch.Tensor, density_k: float = 0.2, eps: float = 1e-8) -> torch.Tensor:
    """color.py:363-434 with gated = gated2 = False (as constructed); ``this_k`` = density_k.item()."""
    pi = 3.141592653589793
    H, V, I = image[:, 0], image[:, 1], image[:, 2]
    H, V, I = torch.clamp(H, -1, 1), torch.clamp(V, -1, 1), torch.clamp(I, 0, 1)
    v = I
    color_sensitive = ((v * 0.5 * pi).sin() + eps).pow(density_k)
    H = torch.clamp(H / (color_sensitive + eps), -1, 1)
    V = torch.clamp(V / (color_sensitive + eps), -1, 1)
    h = (torch.atan2(V, H) / (2 * pi)) % 1
    s = torch.clamp(torch.sqrt(H ** 2 + V ** 2), 0, 1)
    v = torch.clamp(v, 0, 1)
    r, g, b = torch.zeros_like(h), torch.zeros_like(h), torch.zeros_like(h)
    hi = torch.floor(h * 6.0)
    f = h * 6.0 - hi
    p, q, t = v * (1. - s), v * (1. - (f * s)), v * (1. - ((1. - f) * s))
    for k, (rr, gg, bb) in enumerate([(v, t, p), (q, v, p), (p, v, t), (p, q, v), (t, p, v), (v, p, q)]):
        m = hi == k
        r[m], g[m], b[m] = rr[m], gg[m], bb[m]
    return torch.stack([r, g, b], dim=1)


def lllinet_hvi_forward(sd: dict, x: torch.Tensor) -> dict:
    """lllinet.py:360-385: the LLLINet body on rgb_to_hvi(x), its (unclamped) output through hvi_to_rgb, clamp to [0, 1]."""
    k = float(sd["trans.density_k"].reshape(-1)[0]) if "trans.density_k" in sd else 0.2
    y = lllinet_forward(sd, rgb_to_hvi(x, k), clamp=False)["enhanced"]
    return {"enhanced": torch.clamp(hvi_to_rgb(y, k), 0, 1), "hvi": rgb_to_hvi(x, k)}


# ---- U-Net ----------------------------------------------------------------------------------------------------
def _bn(sd, n, x):
    return F.batch_norm(x, sd[f"{n}.running_mean"], sd[f"{n}.running_var"], sd[f"{n}.weight"], sd[f"{n}.bias"], False, 0.1, 1e-5)


def _dc(sd, n, x):
    x = F.relu(_bn(sd, f"{n}.norm1", _c(sd, f"{n}.conv1", x)))
    return F.relu(_bn(sd, f"{n}.norm2", _c(sd, f"{n}.conv2", x)))


def unet_from_mtfe_keys(sd: dict) -> dict:
    """Checkpoint keys of the original U-Net modules (mon_extra/.../mtfe/unet_parts.py:8-77: ``double_conv`` Sequential with
    conv / BN at indices 0,1,3,4, ``downK.maxpool_conv.1``) -> mon's spelling (unet.py:42-47,66: conv1/norm1/conv2/norm2, ``downK.conv``)."""
    idx = {"0": "conv1", "1": "norm1", "3": "conv2", "4": "norm2"}
    out = {}
    for k, v in sd.items():
        if k.endswith("num_batches_tracked"):
            continue
        parts = k.replace(".maxpool_conv.1.", ".conv.").split(".")
        if "double_conv" in parts:
            i = parts.index("double_conv")
            parts[i:i + 2] = [idx[parts[i + 1]]]
        out[".".join(parts)] = v
    return out


def unet_forward(sd: dict, x: torch.Tensor, upsampler: str = "bilinear", enhance: bool = False, taps: bool = False) -> dict:
    """unet.py:172-197.  ``upsampler``: "bilinear" (unet.py:82-83), "convtranspose" (unet.py:85-86, the ``bilinear=False`` branch:
    ``upK.up`` = ConvTranspose2d(in, in // 2, 2, 2)), or "pixelshuffle" (edsr.py:88-123).  Convs may carry a bias (the original modules do)."""
    x1 = _dc(sd, "inc", x)
    x2 = _dc(sd, "down1.conv", F.max_pool2d(x1, 2))
    x3 = _dc(sd, "down2.conv", F.max_pool2d(x2, 2))
    x4 = _dc(sd, "down3.conv", F.max_pool2d(x3, 2))
    x5 = _dc(sd, "down4.conv", F.max_pool2d(x4, 2))

    def up(k, a, skip):
        if upsampler == "bilinear":
            a = F.interpolate(a, scale_factor=2, mode="bilinear", align_corners=True)
        elif upsampler == "convtranspose":
            a = F.conv_transpose2d(a, sd[f"up{k}.up.weight"], sd.get(f"up{k}.up.bias"), stride=2)
        else:
            a = F.pixel_shuffle(_c(sd, f"up{k}.up.0", a), 2)
        dy, dx = skip.shape[2] - a.shape[2], skip.shape[3] - a.shape[3]
        a = F.pad(a, [dx // 2, dx - dx // 2, dy // 2, dy - dy // 2])
        return _dc(sd, f"up{k}.conv", torch.cat([skip, a], 1))
    y = up(1, x5, x4); y = up(2, y, x3); y = up(3, y, x2); y = up(4, y, x1)
    y = _c(sd, "outc.conv", y, padding=0)
    out = {"enhanced": torch.clamp(y, 0, 1)} if enhance else {"semantic": y}
    if taps:
        out.update(inc=x1, bottleneck=x5)
    return out


# ---- Zero-IG Enhance -------------------------------------------------------------------------------------------
def zero_ig_enhance_forward(sd: dict, x: torch.Tensor, layers: int = 3) -> dict:
    fea = F.relu(_c(sd, "in_conv.0", x))
    for _ in range(layers):
        fea = fea + F.relu(_bn(sd, "conv.1", _c(sd, "conv.0", fea)))
    fea = torch.sigmoid(_c(sd, "out_conv.0", fea))
    return {"enhanced": torch.clamp(fea, 0.0001, 1)}


def zero_ig_denoise_forward(sd: dict, x: torch.Tensor, prefix: str) -> torch.Tensor:
    """Denoise1 / Denoise2 (zero_ig.py:257-304): Conv3x3 + LeakyReLU(0.2), Conv3x3 + LeakyReLU(0.2), Conv1x1."""
    y = F.leaky_relu(_c(sd, f"{prefix}.conv1", x), 0.2)
    y = F.leaky_relu(_c(sd, f"{prefix}.conv2", y), 0.2)
    return _c(sd, f"{prefix}.conv3", y, padding=0)


def zero_ig_forward(sd: dict, x: torch.Tensor, layers: int = 3) -> dict:
    """ZeroIG.forward, predicting branch (zero_ig.py:434-452; == Finetunemodel.forward, mon_extra/.../zero_ig/model.py:188-199).
    Keys: ``enhance.*``, ``denoise1.*``, ``denoise2.*``."""
    eps = 1e-4
    image = x + eps
    l2 = torch.clamp(image - zero_ig_denoise_forward(sd, image, "denoise1"), eps, 1.0)
    s2 = zero_ig_enhance_forward({k[8:]: v for k, v in sd.items() if k.startswith("enhance.")}, l2, layers)["enhanced"]
    h2 = torch.clamp(image / s2, eps, 1.0)
    hs = torch.cat([h2, s2], 1)
    h5 = torch.clamp(hs - zero_ig_denoise_forward(sd, hs, "denoise2"), eps, 1.0)
    return {"denoise": h5[:, :3], "enhanced": h2}


# ---- RRDNet / ZS-N2N (parity unpinned: the mon package does not import here; leaf ops are torch's) ---
def rrdnet_forward(sd: dict, x: torch.Tensor, gamma: float = 0.4) -> dict:
    """src/mon/vision/enhance/llie/rrdnet/rrdnet.py:208-240 (three Conv3x3 stacks), :271-289 (forward)."""
    def net(n):
        y = x
        for i in (0, 2, 4, 6):
            y = F.relu(_c(sd, f"{n}.{i}", y))
        return _c(sd, f"{n}.8", y)
    illumination = torch.sigmoid(net("illumination_net"))
    reflectance = torch.sigmoid(net("reflectance_net"))
    noise = torch.tanh(net("noise_net"))
    adjust = torch.pow(illumination, gamma)
    enhanced = torch.clamp(adjust * ((x - noise) / illumination), min=0, max=1)
    return {"image": x, "illumination": illumination, "reflectance": reflectance, "noise": noise, "enhanced": enhanced}


def zsn2n_forward(sd: dict, x: torch.Tensor, predicting: bool = True) -> dict:
    """src/mon/vision/enhance/denoise/zsn2n/zsn2n.py:73-76, 116-124."""
    y = F.leaky_relu(_c(sd, "conv1", x), 0.2)
    y = F.leaky_relu(_c(sd, "conv2", y), 0.2)
    y = _c(sd, "conv3", y, padding=0)
    return {"enhanced": torch.clamp(y, 0, 1) if predicting else y}


# ---- RetinexNet ------------------------------------------------------------------------------------------------
def retinex_decom(sd: dict, x: torch.Tensor):
    """DecomNet (retinexnet/model.py:21-51): returns (R, L)."""
    R = "replicate"
    xin = torch.cat((torch.max(x, dim=1, keepdim=True)[0], x), dim=1)
    f = _c(sd, "net1_conv0", xin, padding=4, mode=R)
    for i in (0, 2, 4, 6, 8):
        f = F.relu(_c(sd, f"net1_convs.{i}", f, mode=R))
    outs = _c(sd, "net1_recon", f, mode=R)
    return torch.sigmoid(outs[:, 0:3]), torch.sigmoid(outs[:, 3:4])


def retinex_relight(sd: dict, i_low: torch.Tensor, r_low: torch.Tensor) -> torch.Tensor:
    """RelightNet (retinexnet/model.py:54-90)."""
    return retinexnet_forward(sd, None, (r_low, i_low))["illumination_delta"]


def retinex_smooth(input_I: torch.Tensor, input_R: torch.Tensor) -> torch.Tensor:
    """RetinexNet.gradient / ave_gradient / smooth (retinexnet/model.py:240-260)."""
    kx = torch.tensor([[0.0, 0.0], [-1.0, 1.0]], dtype=input_I.dtype).view(1, 1, 2, 2)
    ky = kx.transpose(2, 3)
    grad = lambda t, k: torch.abs(F.conv2d(t, k, stride=1, padding=1))
    ave = lambda t, k: F.avg_pool2d(grad(t, k), kernel_size=3, stride=1, padding=1)
    g = (0.299 * input_R[:, 0] + 0.587 * input_R[:, 1] + 0.114 * input_R[:, 2]).unsqueeze(1)
    return torch.mean(grad(input_I, kx) * torch.exp(-10 * ave(g, kx)) + grad(input_I, ky) * torch.exp(-10 * ave(g, ky)))


def retinexnet_losses(sd: dict, low: torch.Tensor, high: torch.Tensor) -> dict:
    """RetinexNet.forward (retinexnet/model.py:199-238): the two losses the training phases minimise."""
    r_low, i_low = retinex_decom(sd, low)
    r_high, i_high = retinex_decom(sd, high)
    i_delta = retinex_relight(sd, i_low, r_low)
    c3 = lambda t: torch.cat((t, t, t), 1)
    recon_low, recon_high = F.l1_loss(r_low * c3(i_low), low), F.l1_loss(r_high * c3(i_high), high)
    mutal_low, mutal_high = F.l1_loss(r_high * c3(i_low), low), F.l1_loss(r_low * c3(i_high), high)
    equal_r = F.l1_loss(r_low, r_high.detach())
    relight = F.l1_loss(r_low * c3(i_delta), high)
    loss_decom = recon_low + recon_high + 0.001 * mutal_low + 0.001 * mutal_high + 0.1 * retinex_smooth(i_low, r_low) \
        + 0.1 * retinex_smooth(i_high, r_high) + 0.01 * equal_r
    loss_relight = relight + 3 * retinex_smooth(i_delta, r_low)
    return {"loss_decom": loss_decom, "loss_relight": loss_relight, "reflectance": r_low, "illumination": i_low, "illumination_delta": i_delta,
            "enhanced": r_low * c3(i_delta)}


def retinexnet_forward(sd: dict, x: torch.Tensor, rl=None) -> dict:
    R = "replicate"
    r_low, i_low = retinex_decom(sd, x) if rl is None else rl
    if x is None:
        x = r_low
    img = torch.cat((r_low, i_low), 1)
    out0 = _c(sd, "net2_conv0_1", img, mode=R)
    out1 = F.relu(_c(sd, "net2_conv1_1", out0, stride=2, mode=R))
    out2 = F.relu(_c(sd, "net2_conv1_2", out1, stride=2, mode=R))
    out3 = F.relu(_c(sd, "net2_conv1_3", out2, stride=2, mode=R))
    sz = lambda t: (t.shape[2], t.shape[3])
    d1 = F.relu(_c(sd, "net2_deconv1_1", torch.cat((F.interpolate(out3, size=sz(out2)), out2), 1), mode=R))
    d2 = F.relu(_c(sd, "net2_deconv1_2", torch.cat((F.interpolate(d1, size=sz(out1)), out1), 1), mode=R))
    d3 = F.relu(_c(sd, "net2_deconv1_3", torch.cat((F.interpolate(d2, size=sz(out0)), out0), 1), mode=R))
    feats = torch.cat((F.interpolate(d1, size=sz(x)), F.interpolate(d2, size=sz(x)), d3), 1)
    fus = _c(sd, "net2_fusion", feats, padding=1, mode=R)
    i_delta = _c(sd, "net2_output", fus, padding=0)
    return {"reflectance": r_low, "illumination": i_low, "illumination_delta": i_delta, "enhanced": r_low * i_delta}


# ---- HINet -------------------------------------------------------------------------------------------------------
def hinet_forward(sd: dict, x: torch.Tensor, depth: int = 5, relu_slope: float = 0.2, in_pos_left: int = 0, in_pos_right: int = 4, q=None) -> dict:
    """src/mon/vision/enhance/multitask/hinet/hinet.py:34-198, 322-362 with hinet.py:80's ``if enc and dec`` read as
    ``enc is not None and dec is not None`` (as shipped it raises on tensors; the original repository's hinet_arch.py:147 has
    exactly that test).  Pinned by tests/golden/hinet_*.npz.  ``q``: storage rounding of the 16-bit training graph
    (oracle.train.storage_rounding) applied wherever the engine stores a tensor between two kernels; None = the plain forward."""
    q, qw = q if q is not None else ((lambda t: t), (lambda w: w))          # (stored activations, weights as the 16-bit kernels read them)
    lr = lambda t: F.leaky_relu(t, relu_slope)
    x = q(x)

    def _c(sd_, name, t, padding=1):                                          # shadows the module helper: 16-bit layers read rounded weights
        return F.conv2d(t, qw(sd_[f"{name}.weight"]), sd_.get(f"{name}.bias"), padding=padding)

    def block(n, t, hin, down, enc=None, dec=None):
        idn = q(_c(sd, f"{n}.identity", t, padding=0))
        y = _c(sd, f"{n}.conv_1", t)
        if hin:
            y = q(y)
            y1, y2 = torch.chunk(y, 2, dim=1)
            y = torch.cat([F.instance_norm(y1, weight=sd[f"{n}.norm.weight"], bias=sd[f"{n}.norm.bias"], eps=1e-5), y2], 1)
        y = q(lr(y))
        y = q(q(lr(_c(sd, f"{n}.conv_2", y))) + idn)
        if enc is not None and dec is not None:
            y = q(q(y + q(_c(sd, f"{n}.csff_enc", enc))) + q(_c(sd, f"{n}.csff_dec", dec)))
        if down:
            return q(F.conv2d(y, sd[f"{n}.downsample.weight"], None, stride=2, padding=1)), y
        return y

    def stage(s, t, encs=None, decs=None):
        ups, outs = [], []
        for i in range(depth):
            hin = in_pos_left <= i <= in_pos_right
            if i + 1 < depth:
                e, d = (encs[i], decs[-i - 1]) if encs is not None else (None, None)
                t, up = block(f"down_path_{s}.{i}", t, hin, True, e, d)
                ups.append(up)
            else:
                t = block(f"down_path_{s}.{i}", t, hin, False)
        for j in range(depth - 1):
            bridge = q(_c(sd, f"skip_conv_{s}.{j}", ups[-j - 1]))
            u = q(F.conv_transpose2d(t, sd[f"up_path_{s}.{j}.up.weight"], sd[f"up_path_{s}.{j}.up.bias"], stride=2))
            t = block(f"up_path_{s}.{j}.conv_block", torch.cat([u, bridge], 1), False, False)
            outs.append(t)
        return t, ups, outs

    x1, encs, decs = stage(1, q(_c(sd, "conv_01", x)))
    f1 = q(_c(sd, "sam12.conv1", x1))
    img = q(q(_c(sd, "sam12.conv2", x1)) + x)
    sam = q(q(f1 * q(torch.sigmoid(_c(sd, "sam12.conv3", img)))) + x1)
    x2 = q(_c(sd, "cat12", torch.cat([q(_c(sd, "conv_02", x)), sam], 1), padding=0))
    x2, _, _ = stage(2, x2, encs, decs)
    return {"stage1": img, "enhanced": q(q(_c(sd, "last", x2)) + x)}


# ---- MPRNet ------------------------------------------------------------------------------------------------------
def mprnet_forward(sd: dict, x: torch.Tensor, num_cab: int = 8, taps: dict = None) -> dict:
    """src/mon/vision/enhance/multitask/mprnet/mprnet.py:33-340 (modules) and :455-548 (three-stage forward with the 4 / 2 / 1 patch
    hierarchy).  The ONE nn.PReLU() instance (mprnet.py:421) is shared by every CAB and appears in the checkpoint under each
    ``...body.1.weight``.  The BLOCKS are PINNED (round 4) by tests/golden/mprnet_c8_*.npz: conv / CAB / SAM / Encoder (plain and CSFF) /
    Decoder / ORSNet executed out of mprnet.py (plain torch.nn classes in a module that imports mon), with block-level taps; the model class
    derives from mon's Lightning base and cannot be defined here, so the order of the block calls in forward() (crops, cats) is a
    transcription in the fixture's generator as it is here.  ``taps``: filled with the intermediate lists the fixture holds."""
    def cv(n, t):
        w = sd[f"{n}.weight"]
        return F.conv2d(t, w, sd.get(f"{n}.bias"), padding=w.shape[2] // 2)

    def cab(n, t):
        y = cv(f"{n}.body.2", F.prelu(cv(f"{n}.body.0", t), sd[f"{n}.body.1.weight"]))
        a = F.adaptive_avg_pool2d(y, 1)
        a = torch.sigmoid(cv(f"{n}.CA.conv_du.2", F.relu(cv(f"{n}.CA.conv_du.0", a))))
        return y * a + t

    def cabs(n, t, k=2):
        for i in range(k):
            t = cab(f"{n}.{i}", t)
        return t

    down = lambda n, t: cv(f"{n}.down.1", F.interpolate(t, scale_factor=0.5, mode="bilinear", align_corners=False))
    up = lambda n, t: cv(f"{n}.up.1", F.interpolate(t, scale_factor=2, mode="bilinear", align_corners=False))

    def encoder(n, t, enc=None, dec=None):
        e1 = cabs(f"{n}.encoder_level1", t)
        if enc and dec:
            e1 = e1 + cv(f"{n}.csff_enc1", enc[0]) + cv(f"{n}.csff_dec1", dec[0])
        e2 = cabs(f"{n}.encoder_level2", down(f"{n}.down12", e1))
        if enc and dec:
            e2 = e2 + cv(f"{n}.csff_enc2", enc[1]) + cv(f"{n}.csff_dec2", dec[1])
        e3 = cabs(f"{n}.encoder_level3", down(f"{n}.down23", e2))
        if enc and dec:
            e3 = e3 + cv(f"{n}.csff_enc3", enc[2]) + cv(f"{n}.csff_dec3", dec[2])
        return [e1, e2, e3]

    def decoder(n, e):
        d3 = cabs(f"{n}.decoder_level3", e[2])
        t = up(f"{n}.up32", d3) + cab(f"{n}.skip_attn2", e[1])
        d2 = cabs(f"{n}.decoder_level2", t)
        t = up(f"{n}.up21", d2) + cab(f"{n}.skip_attn1", e[0])
        return [cabs(f"{n}.decoder_level1", t), d2, d3]

    def sam(n, t, img):
        x1 = cv(f"{n}.conv1", t)
        im = cv(f"{n}.conv2", t) + img
        return x1 * torch.sigmoid(cv(f"{n}.conv3", im)) + t, im

    def orb(n, t):
        return cv(f"{n}.body.{num_cab}", cabs(f"{n}.body", t, num_cab)) + t

    shallow = lambda k, t: cab(f"shallow_feat{k}.1", cv(f"shallow_feat{k}.0", t))
    h, w = x.shape[2], x.shape[3]
    top, bot = x[:, :, :h // 2], x[:, :, h // 2:]
    quads = [top[:, :, :, :w // 2], top[:, :, :, w // 2:], bot[:, :, :, :w // 2], bot[:, :, :, w // 2:]]
    f1 = [encoder("stage1_encoder", shallow(1, q)) for q in quads]
    f1_top = [torch.cat((a, b), 3) for a, b in zip(f1[0], f1[1])]
    f1_bot = [torch.cat((a, b), 3) for a, b in zip(f1[2], f1[3])]
    r1_top, r1_bot = decoder("stage1_decoder", f1_top), decoder("stage1_decoder", f1_bot)
    s_top, img_top = sam("sam12", r1_top[0], top)
    s_bot, img_bot = sam("sam12", r1_bot[0], bot)
    stage1 = torch.cat([img_top, img_bot], 2)
    if taps is not None:
        taps.update(feat1_ltop=f1[0], res1_top=r1_top, sam12_top_feats=s_top, sam12_top_img=img_top)
    c_top = cv("concat12", torch.cat([shallow(2, top), s_top], 1))
    c_bot = cv("concat12", torch.cat([shallow(2, bot), s_bot], 1))
    f2_top = encoder("stage2_encoder", c_top, f1_top, r1_top)
    f2_bot = encoder("stage2_encoder", c_bot, f1_bot, r1_bot)
    f2 = [torch.cat((a, b), 2) for a, b in zip(f2_top, f2_bot)]
    r2 = decoder("stage2_decoder", f2)
    s3, stage2 = sam("sam23", r2[0], x)
    t = cv("concat23", torch.cat([shallow(3, x), s3], 1))
    o = "stage3_orsnet"
    t = orb(f"{o}.orb1", t) + cv(f"{o}.conv_enc1", f2[0]) + cv(f"{o}.conv_dec1", r2[0])
    t = orb(f"{o}.orb2", t) + cv(f"{o}.conv_enc2", up(f"{o}.up_enc1", f2[1])) + cv(f"{o}.conv_dec2", up(f"{o}.up_dec1", r2[1]))
    t = orb(f"{o}.orb3", t) + cv(f"{o}.conv_enc3", up(f"{o}.up_enc2.1", up(f"{o}.up_enc2.0", f2[2]))) \
        + cv(f"{o}.conv_dec3", up(f"{o}.up_dec2.1", up(f"{o}.up_dec2.0", r2[2])))
    if taps is not None:
        taps.update(feat2_top=f2_top, res2=r2, orsnet=t)
    return {"stage1": stage1, "stage2": stage2, "enhanced": cv("tail", t) + x}
